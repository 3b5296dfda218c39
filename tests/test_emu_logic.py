"""Kernel-LOGIC checks on the CPU-only container (TEST INFRASTRUCTURE, not a product path).

tests/emu builds the unmodified HIP sources against a host-thread emulation of the few HIP constructs they use
(tests/emu/hip/hip_runtime.h) and these tests run the same parity cases as tests/test_gpu_parity.py through it, against
the oracle.  They catch indexing / barrier / bookkeeping mistakes before GPU minutes are spent; they say nothing about
the device (the `-m gpu` tests are the parity tests proper) and the emulated library is never importable from
cubicsdr_amd/ (checked by tests/test_abi.py).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import tests.test_gpu_parity as G

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

# the quick subset (~2 min on 8 cores) runs by default; CSDR_EMU_FULL=1 runs every case (add CSDR_EMU_FLAVOR=asan|tsan and
# LD_PRELOAD=$(gcc -print-file-name=libasan.so|libtsan.so) for the sanitizer builds)
full = pytest.mark.skipif(os.environ.get("CSDR_EMU_FULL", "0") != "1", reason="set CSDR_EMU_FULL=1 for the full emulation suite")


@pytest.fixture(scope="module")
def ctx():
    import build_emu
    import cubicsdr_amd.hip as H
    from cubicsdr_amd.engine import Context
    path = build_emu.build(os.environ.get("CSDR_EMU_FLAVOR", ""))
    lib = C.CDLL(path)
    for name, (res, args) in H.ABI.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    saved = H._lib
    H._lib = lib            # the engine objects created inside this module talk to the emulated library
    c = Context(0)
    try:
        yield c
    finally:
        c.close()
        H._lib = saved


def test_emu_channelizer_c1(ctx):
    G.test_channelizer_matches_firpfbch(ctx, 2400000, 4, 40000)


@full
def test_emu_channelizer_m6(ctx):
    G.test_channelizer_matches_firpfbch(ctx, 3000000, 6, 50004)


@pytest.mark.parametrize("fs,M,block", [pytest.param(2400000, 4, 40000, marks=full), (6100000, 122, 122 * 70), (6100000, 122, 122 * 3)])
def test_emu_channelizer_batched(ctx, fs, M, block):
    G.test_channelizer_batched_equals_blockwise(ctx, fs, M, block)


def test_emu_channelizer2_m6(ctx):
    G.test_channelizer2_matches_firpfbch2(ctx, 3000000, 6, 5004)


@full
def test_emu_channelizer2_m20(ctx):
    G.test_channelizer2_matches_firpfbch2(ctx, 10000000, 20, 16680)


@full
def test_emu_demods_behind_oversampled_channelizer(ctx):
    G.test_demods_behind_oversampled_channelizer(ctx)


def test_emu_dc_blocker(ctx):
    G.test_single_channel_dc_blocker(ctx)


def test_emu_nbfm_c1(ctx):
    G.test_nbfm_c1_config(ctx)


def test_emu_depth_two_cascade(ctx):
    G.test_nbfm_from_100k_channels_depth_two_cascade(ctx)


@full
def test_emu_mixed_modems(ctx):
    G.test_mixed_modems_streaming(ctx)


@full
def test_emu_tiny_blocks(ctx):
    G.test_tiny_blocks_ragged_outputs(ctx)


@full
def test_emu_dsb(ctx):
    G.test_dsb_modem_costas_loop(ctx)


@full
def test_emu_cw_ten_stage(ctx):
    G.test_cw_from_a_wide_channel_ten_stage_cascade(ctx)


@full
def test_emu_cw(ctx):
    G.test_cw_modem(ctx)


@full
def test_emu_iq_passthrough(ctx):
    G.test_iq_passthrough_modem(ctx)


@full
def test_emu_wide_fm(ctx):
    G.test_wide_fm_audio_decimation(ctx)


@full
def test_emu_batched(ctx):
    G.test_batched_equals_reference(ctx)


@full
def test_emu_single_channel_demod(ctx):
    G.test_single_channel_mode_demod(ctx)


@pytest.mark.parametrize("F", [pytest.param(512, marks=full), 2048, pytest.param(16384, marks=full)])
def test_emu_fft(ctx, F):
    G.test_fft_matches_liquid(ctx, F)


@full
def test_emu_spectrum_first_frame(ctx):
    G.test_spectrum_points_first_frame_mode(ctx, 2048, 40000)


def test_emu_spectrum_contiguous(ctx):
    G.test_spectrum_contiguous_mode(ctx)


def test_emu_spectrum_line_cadence(ctx):
    G.test_spectrum_line_cadence_overlapped_frames(ctx, 1024, 600)


@full
def test_emu_spectrum_line_cadence_half_frame_lines(ctx):
    G.test_spectrum_line_cadence_overlapped_frames(ctx, 2048, 2048)


@full
def test_emu_spectrum_peak_hold_hide_dc(ctx):
    G.test_spectrum_peak_hold_and_hide_dc(ctx)


def test_emu_spectrum_zoomed_view(ctx):
    G.test_spectrum_zoomed_view(ctx)


def test_emu_spectrum_zoomed_view_not_a_power_of_two(ctx):
    G.test_spectrum_zoomed_view(ctx, 600)


def test_emu_spectrum_zoomed_view_two_pass(ctx):
    G.test_spectrum_zoomed_view_behind_a_two_pass_transform(ctx)


@full
def test_emu_spectrum_many_frames(ctx):
    G.test_spectrum_many_frames_one_batch(ctx)


def test_emu_routing_follows_retunes(ctx):
    G.test_routing_follows_centre_and_demodulator_retunes(ctx)


@full
def test_emu_retune_skip_inactive(ctx):
    G.test_retune_skip_and_inactive(ctx)


def test_emu_error_codes(ctx):
    G.test_error_codes_and_edge_inputs(ctx)


@full
def test_emu_interpolating_iq_resampler(ctx):
    got, want = G._run_demods(ctx, 2400000, 4, 4000, ["FM", "NBFM"], 4, 2, bw=[800000, 12500], seed=21)
    print(G._compare(got, want, "interp"))


def test_emu_fm_stereo(ctx):
    print(G._fms_case(ctx, 2400000, 4, 20000, 4, 2))


def test_emu_time_slab_sharding(ctx):
    print(G._slab_case(ctx, 480000, 8, 8000, 6, 2, 4, 3, False, kinds=("NBFM", "AM")))
@full
def test_emu_fm_stereo_settings(ctx):
    print(G._fms_case(ctx, 2400000, 4, 20000, 4, 2, bw=150000, audio_rate=44100, demph=0, seed=37))
    print(G._fms_case(ctx, 2400000, 4, 20000, 4, 2, bw=50000, audio_rate=48000, demph=50, seed=37))


@pytest.mark.parametrize("fs,M,block", [(5000000, 10, 10 * 130), pytest.param(7000000, 14, 14 * 70, marks=full), (6100000, 122, 122 * 150), (5900000, 118, 118 * 70),
                                        pytest.param(3700000, 74, 74 * 80, marks=full)])
def test_emu_channelizer_m_twice_odd(ctx, fs, M, block):
    """M = 2 A, A odd: the one-lane-per-frame kernel with its mover wave (ragged and whole 64-frame tiles, carried history)"""
    G.test_channelizer_m_twice_odd(ctx, fs, M, block)


@pytest.mark.parametrize("M,frames", [pytest.param(M, fr, marks=() if M in (4, 20, 200, 56, 32, 68, 202, 116) and fr != 3 else full) for M, fr in G.FFT_SIZES])
def test_emu_channelizer_fft_sizes(ctx, M, frames):
    """the mixed-radix FFT channelizer (kernels_chanfft.hpp): tile walk, FIR windows across the history, every pass's indexing"""
    G.test_channelizer_fft_sizes(ctx, M, frames)


@pytest.mark.parametrize("fs,M,block", [(4000000, 8, 8 * 77), pytest.param(20000000, 40, 40 * 61, marks=full), pytest.param(34000000, 68, 68 * 45, marks=full),
                                        (6100000, 122, 122 * 77), (1900000, 38, 38 * 100), pytest.param(6300000, 126, 126 * 9, marks=full), pytest.param(1700000, 34, 34 * 130, marks=full)])
def test_emu_channelizer2_fft(ctx, fs, M, block):
    """firpfbch2 inside the FFT channelizer (round 5): the two lattices of frames, windows across the 7.5 M history, post factors and the sign of odd channels"""
    G.test_channelizer2_matches_firpfbch2(ctx, fs, M, block)


@full
def test_emu_spectrum_headline_fused_chain(ctx):
    """the two-pass chain of the headline size (kernels_spec3.hpp, round 5): 1024-thread column workgroups walking frames, the row pass's producer /
    consumer waves over a partial round, a frame carried from call to call -- five frames of 2^17 points over three calls against the reference
    (seven minutes of host threads: 256 row-pair workgroups of 768, 16 column workgroups of 1024; part of the full suite only)"""
    G._spectrum_contiguous_batches(ctx, 65536, 61440000, (2, 2, 1))


def test_emu_spectrum_contiguous_batches_multi_row(ctx):
    """the control flow of the headline spectrum test (carry across calls, > 256-frame rounds, row-pair tiles) at a size the emulation finishes"""
    G._spectrum_contiguous_batches(ctx, 4096, 2400000, (8, 20, 5))


@full
def test_emu_spectrum_contiguous_batches_multi_row_rounds(ctx):
    G._spectrum_contiguous_batches(ctx, 4096, 2400000, (40, 270, 33))


def test_emu_scope_against_reference(ctx):
    import tests.test_gpu_io as IO
    IO.scope_scenario(ctx)


def test_emu_mixer_against_reference_callback(ctx):
    import tests.test_gpu_io as IO
    IO.mixer_scenario(ctx, n_callbacks=70)


def test_emu_ingest_shared_device_buffer(ctx):
    import tests.test_gpu_io as IO
    IO.ingest_scenario(ctx, block=8000, rounds=3, F=512)


def test_emu_spectrum_nan_repairs_frame_by_frame(ctx):
    G.test_spectrum_nan_samples_recover_frame_by_frame(ctx, 2048, (40, 23), ((5, 100), (38, 7), (41, 3000)))


def test_emu_comm_refusals(ctx):
    G.test_comm_refuses_bad_arguments(ctx)


def test_emu_comm_one_rank(ctx):
    """csdr_comm's entry points and the drivers' ABI transport with the one-rank loopback of the host-executing build"""
    G._comm_one_rank_case(ctx, False)


def test_emu_slab_exchange_overlapped(ctx):
    """the two-halves row exchange (begin / finish, the producer's rotating buffers, the two receive slots) against the one-call form: five batches"""
    G._slab_overlap_case(False)


def test_emu_cpp_comm_ranks(ctx):
    """tests/cpp/comm_ranks.cpp (the C++ host of a sharded stream, C ABI only) linked against the host-executing build: one rank, loopback"""
    import subprocess
    import build_emu
    lib = build_emu.build(os.environ.get("CSDR_EMU_FLAVOR", ""))
    exe = os.path.join(os.path.dirname(lib), "comm_ranks_emu")
    src = os.path.join(HERE, "cpp", "comm_ranks.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-pthread", src, "-o", exe, "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib), "-ldl",
                    "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded rows: ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("M", [10, 20, 68])
def test_emu_packed_row_order(ctx, M):
    G.test_packed_row_order_holds_the_same_rows(ctx, M)


def test_emu_spectrum_sizes_that_are_not_powers_of_two(ctx):
    for F in (600, 37, 3, 1500):
        G.test_fft_matches_liquid(ctx, F)
    G.test_spectrum_points_first_frame_mode(ctx, 375, 4000)
    G._spectrum_contiguous_batches(ctx, 375, 2400000, (5, 3, 4))
    G._spectrum_contiguous_batches(ctx, 1500, 2400000, (5, 3, 4))
