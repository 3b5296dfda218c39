"""Parity of the path's edges on the GPU (through the C ABI) against the reference's OWN classes compiled unmodified into oracle/_ref:
ScopeVisualProcessor (libref_scope.so), audioCallback + AudioFileWAV (libref_audio.so).  The scenario bodies are plain functions so that
tests/test_emu_logic.py can run them through the host-thread emulation on the CPU-only container."""
import os
import struct
import tempfile

import numpy as np
import pytest

from tests.util import demod_frequencies, rel_err, synth_iq

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    from cubicsdr_amd.engine import Context
    c = Context(0)
    yield c
    c.close()


def _need(flag, what):
    if not flag:
        pytest.skip("oracle/_ref/%s is built only where /root/reference is (it travels to the GPU box prebuilt)" % what)


# ----------------------------------------------------------------------------------------------- audio scope
def _scope_frames(rng):
    """a sequence of AudioThreadInputs as DemodulatorThread hands them to the scope (DemodulatorThread.cpp:240-316)"""
    t = np.arange(4096)
    tone = lambda f, a, n: (a * np.sin(2 * np.pi * f * t[:n] / 48000.0) + 0.01 * rng.standard_normal(n)).astype(np.float32)
    fr = []
    fr.append(dict(data=tone(1000, 0.4, 800), channels=1, type=0, sample_rate=12500, input_rate=48000))     # NBFM: audio tap, bandwidth label below the audio rate
    fr.append(dict(data=tone(1000, 0.4, 801), channels=1, type=0, sample_rate=12500, input_rate=48000))
    fr.append(dict(data=tone(700, 2.5, 90), channels=1, type=0, sample_rate=5400, input_rate=5400))         # USB: demodulator-output tap, peak above 1
    fr.append(dict(data=tone(3000, 0.2, 2048), channels=1, type=0, sample_rate=200000, input_rate=200000))  # more samples than the scope shows
    st = np.concatenate([tone(400, 0.3, 800), tone(900, 0.6, 800)])                                          # stereo, planar halves
    fr.append(dict(data=st, channels=2, type=1, sample_rate=36000, input_rate=48000))
    fr.append(dict(data=(st * 1.7).astype(np.float32), channels=2, type=1, sample_rate=48000, input_rate=48000))
    xy = np.empty(1600, np.float32); xy[0::2] = tone(500, 0.5, 800); xy[1::2] = tone(500, 0.5, 800)[::-1]
    fr.append(dict(data=xy, channels=2, type=2, sample_rate=48000, input_rate=48000))                        # X / Y pairs
    for k in range(9):
        fr.append(dict(data=tone(1000 + 150 * k, 0.1 + 0.1 * k, 780 + 7 * k), channels=1, type=0, sample_rate=12500, input_rate=48000))
    return fr


def _compare_scope_item(got, want, tag):
    assert got is not None, tag
    for k in ("mode", "spectrum", "channels", "input_rate", "sample_rate"):
        assert got[k] == want[k], (tag, k, got[k], want[k])
    assert got["points"].size == want["points"].size, (tag, got["points"].size, want["points"].size)
    if want["spectrum"]:
        assert got["fft_size"] == want["fft_size"], tag
        assert abs(got["fft_floor"] - want["fft_floor"]) <= TOL * abs(want["fft_ceil"]) and abs(got["fft_ceil"] - want["fft_ceil"]) <= TOL * abs(want["fft_ceil"]), tag
        assert np.array_equal(got["points"][0::2], want["points"][0::2]), tag                    # the x positions are exact
        return rel_err(got["points"][1::2], want["points"][1::2])
    # waveform: x positions and the normalised samples are individually rounded operations -> bit for bit
    assert np.array_equal(got["points"], want["points"]), (tag, float(np.max(np.abs(got["points"] - want["points"]))))
    return 0.0


def scope_scenario(ctx, per_call=(1, 1, 3, 2, 5, 4), fft_size=1024):
    import oracle.ref_modems as RM
    from cubicsdr_amd.engine import ScopeProcessor
    _need(RM.scope_available(), "libref_scope.so")
    rng = np.random.default_rng(5)
    frames = _scope_frames(rng)
    ref = RM.RefScopeCpp(fft_size)
    sp = ScopeProcessor(ctx, fft_size, max_frames=8, max_samples=4096)
    worst = 0.0
    k = 0
    for n in per_call:
        batch = frames[k:k + n]
        sp.process(batch)
        for j, f in enumerate(batch):
            want = ref.push(f["data"], f["channels"], f["input_rate"], f["sample_rate"], f["type"])
            assert len(want) == 2 and not want[0]["spectrum"] and want[1]["spectrum"]
            worst = max(worst, _compare_scope_item(sp.fetch(j, False), want[0], (k + j, "wave")))
            e = _compare_scope_item(sp.fetch(j, True), want[1], (k + j, "spectrum"))
            assert e < TOL, (k + j, e)
            worst = max(worst, e)
        k += n
    assert k == len(frames)
    # spectrum only, then scope only (setScopeEnabled / setSpectrumEnabled)
    sp.set_enabled(False, True); ref.enable(False, True)
    sp.process(frames[:1])
    want = ref.push(frames[0]["data"], 1, 48000, 12500, 0)
    assert len(want) == 1 and sp.fetch(0, False) is None
    assert _compare_scope_item(sp.fetch(0, True), want[0], "spectrum only") < TOL
    sp.set_enabled(True, False); ref.enable(True, False)
    sp.process(frames[1:2])
    want = ref.push(frames[1]["data"], 1, 48000, 12500, 0)
    assert len(want) == 1 and sp.fetch(0, True) is None
    _compare_scope_item(sp.fetch(0, False), want[0], "scope only")
    sp.close(); ref.close()
    return worst


@pytest.mark.parametrize("fft_size", [1024, 256, 4096])
def test_scope_matches_reference_processor(ctx, fft_size):
    """waveform items bit for bit, spectrum items at 1e-5, floor / ceil trackers, item sizes (decimated taps), over 16 frames in calls of 1..5;
    fftSize 4096 is the largest csdr_scope_setup accepts (its LDS request passes the 64 KB default)"""
    print("audio scope (fftSize %d) against the reference's ScopeVisualProcessor: worst %.3g" % (fft_size, scope_scenario(ctx, fft_size=fft_size)))


def test_scope_tap_is_the_modems_demod_output(ctx):
    """csdr_bank_fetch_demod_output = ModemAnalog::getDemodOutputData of the last block (the gain-scaled demodulator output in front of the audio
    resampler, whole block, at most DEMOD_VIS_SIZE samples): compared with the reference's OWN modem classes (libref_modems.so) fed the same
    resampled IQ block by block -- NBFM, AM at 6 kHz (interpolating audio), AM at 100 kHz and FM at 200 kHz (decimating audio cascades, whose
    trailing samples of a block are only consumed by the NEXT block's outputs), USB."""
    import oracle.ref_modems as RM
    from cubicsdr_amd.engine import DemodBank, SDRPost
    _need(RM.available(), "libref_modems.so")
    fs, M, block, center = 2400000, 4, 40000, 100000000
    kinds = [("NBFM", 12500), ("AM", 6000), ("AM", 100000), ("FM", 200000), ("USB", 5400)]
    freqs = demod_frequencies(center, fs, len(kinds))
    post = SDRPost(ctx, fs, M, block, max_blocks=1)
    bank = DemodBank(ctx, len(kinds), max_blocks=1)
    refs = [RM.RefModem(k, bw) for k, bw in kinds]
    for i, ((k, bw), f) in enumerate(zip(kinds, freqs)):
        bank.configure(i, post, k, bw, f)
    worst = 0.0
    for b in range(4):
        x = synth_iq(block, fs, center, [(k, f) for (k, _), f in zip(kinds, freqs)], seed=300 + b, t0=b * block)
        post.execute(x, 1, block, center)
        bank.execute(post)
        for i, (k, bw) in enumerate(kinds):
            iq = bank.iq(i)
            refs[i].demodulate(iq)
            want = refs[i].demod_output()[:2048]
            got = bank.demod_output(i)
            assert got.size == want.size == min(iq.size, 2048), (b, k, bw, got.size, want.size, iq.size)
            e = rel_err(got, want)
            assert e < TOL, (b, k, bw, e)
            worst = max(worst, e)
    print("scope tap against the reference modems' demodOutputData: worst %.3g" % worst)
    for r in refs:
        r.close()
    bank.close(); post.close()


def test_scope_taps_read_in_place_from_the_bank(ctx):
    """csdr_bank_scope_frame: the tap of DemodulatorThread.cpp:240-316 as a device-resident frame -- NBFM and USB (audio tap: more audio than
    IQ samples), AM at 100 kHz (demodulator-output tap), I/Q (0.75 x re | im), FM stereo (left | right, labelled 36000) -- through csdr_scope
    on the device, against the reference scope fed the AudioThreadInput the reference's statements would have assembled from the fetched
    audio / tap (the tap's content itself: test_scope_tap_is_the_modems_demod_output)."""
    import oracle.ref_modems as RM
    from cubicsdr_amd.engine import DemodBank, ScopeProcessor, SDRPost
    _need(RM.scope_available(), "libref_scope.so")
    fs, M, block, center = 2400000, 4, 40000, 100000000
    kinds = [("NBFM", 12500), ("AM", 100000), ("I/Q", 48000), ("FMS", 200000), ("USB", 5400)]
    freqs = demod_frequencies(center, fs, len(kinds))
    post = SDRPost(ctx, fs, M, block, max_blocks=2)
    bank = DemodBank(ctx, len(kinds), max_blocks=2)
    for i, ((k, bw), f) in enumerate(zip(kinds, freqs)):
        bank.configure(i, post, k, bw, f)
    sp = ScopeProcessor(ctx, 1024, max_frames=len(kinds), max_samples=8192)
    ref = RM.RefScopeCpp(1024)
    worst = 0.0
    for rnd in range(3):
        x = synth_iq(2 * block, fs, center, [("NBFM", freqs[0]), ("AM", freqs[1]), ("NBFM", freqs[2]), ("FMS", freqs[3]), ("USB", freqs[4])], seed=70 + rnd, t0=rnd * 2 * block)
        post.execute(x, 2, block, center)
        bank.execute(post)
        frames = [bank.scope_frame(i) for i in range(len(kinds))]
        assert all(f.n > 0 for f in frames)
        sp.process(frames)
        for i, (k, bw) in enumerate(kinds):
            r = bank.results(i)[-1]
            audio = bank.audio(i)[r.audio_offset:r.audio_offset + r.n_audio]
            # the AudioThreadInput the reference assembles (:254-312)
            if k == "I/Q":
                n = min(audio.size, 4096)
                data = np.concatenate([audio[1:n:2] * np.float32(0.75), audio[0:n:2] * np.float32(0.75)]); ch, typ, sr, ir = 2, 1, bw, bw
            elif k == "FMS":
                n = min(audio.size, 4096)
                data = np.concatenate([audio[0:n:2], audio[1:n:2]]); ch, typ, sr, ir = 2, 1, 36000, 48000
            elif r.n_audio > r.n_iq:
                data = audio[:2048]; ch, typ, sr, ir = 1, 0, bw, 48000
            else:
                data = bank.demod_output(i)[:2048]; ch, typ, sr, ir = 1, 0, bw, bw
                assert data.size == min(r.n_iq, 2048)
            want = ref.push(data, ch, ir, sr, typ)
            _compare_scope_item(sp.fetch(i, False), want[0], (rnd, k, "wave"))
            e = _compare_scope_item(sp.fetch(i, True), want[1], (rnd, k, "spectrum"))
            assert e < TOL, (rnd, k, e)
            worst = max(worst, e)
    print("scope taps read in HBM: worst spectrum error %.3g" % worst)
    sp.close(); ref.close(); bank.close(); post.close()


# ----------------------------------------------------------------------------------------------- audio mix-down
def mixer_scenario(ctx, n_callbacks=90, frames=256):
    """five sources through ~90 callbacks: mono and stereo, gains, ragged block sizes that straddle callbacks, a loud stretch (normalisation), blocks
    at another sample rate, an empty block, a source switched inactive and back, a source that runs dry and resumes, a bounded queue that drops"""
    import oracle.ref_modems as RM
    from cubicsdr_amd.engine import AudioMixer
    _need(RM.audio_available(), "libref_audio.so")
    rng = np.random.default_rng(11)
    NS, RATE, QCAP = 5, 48000, 6
    ref = RM.RefAudioMixCpp(RATE, NS, QCAP)
    mix = AudioMixer(ctx, NS, RATE, ring_floats=1 << 16, queue_blocks=QCAP)
    gains = [1.0, 0.5, 0.8, 1.3, 0.25]
    for i, g in enumerate(gains):
        ref.set_source(i, True, g); mix.set_source(i, True, True, g, QCAP)
    channels = [1, 2, 1, 1, 2]

    def push(i, n, amp, rate=RATE, ch=None):
        ch = channels[i] if ch is None else ch
        data = (amp * rng.standard_normal(n * max(ch, 1))).astype(np.float32)
        peak = float(np.max(np.abs(data))) if data.size else 0.0
        a = ref.push(i, data, ch, rate, peak)
        b = mix.push(i, data, ch, rate, peak)
        assert a == b, (i, a, b)

    exact = 0
    for cb in range(n_callbacks):
        # producers: roughly one block per source per callback, ragged sizes; scripted irregularities
        for i in range(NS):
            if i == 2 and 30 <= cb < 40:
                continue                                                      # source 2 runs dry, then resumes
            if i == 3 and cb == 12:
                push(i, 200, 0.2, rate=44100); push(i, 210, 0.2, rate=44100)   # blocks at another rate are discarded
            if i == 0 and cb == 20:
                push(i, 0, 0.0)                                               # an empty block
            amp = 0.9 if 50 <= cb < 60 else 0.15                              # a loud stretch: the summed peaks exceed 1
            n = int(frames * (0.6 + 0.9 * rng.random()))
            push(i, n, amp)
            if i == 4 and cb % 7 == 0:
                push(i, 40, 0.1); push(i, 33, 0.1); push(i, 500, 0.1)         # bursts: the bounded queue drops some
        if cb == 25:
            ref.set_source(1, False, gains[1]); mix.set_source(1, True, False, gains[1], QCAP)
        if cb == 33:
            ref.set_source(1, True, 0.7); mix.set_source(1, True, True, 0.7, QCAP)
        want = ref.callback(frames)
        got = mix.render(frames, 1)
        assert np.array_equal(got, want), (cb, float(np.max(np.abs(got - want))))
        for i in range(NS):
            assert mix.queued(i) == ref.queued(i), (cb, i)
        exact += 1
    # several callbacks rendered in ONE call equal the same callbacks one at a time
    for i in range(NS):
        for _ in range(4):
            push(i, int(frames * 1.1), 0.2)
    want = np.concatenate([ref.callback(frames) for _ in range(3)])
    got = mix.render(frames, 3)
    assert np.array_equal(got, want)
    mix.close(); ref.close()
    return exact


def test_mixer_is_the_reference_callback_bit_for_bit(ctx):
    print("mixer: %d callbacks identical to the reference's audioCallback" % mixer_scenario(ctx))


def test_mixer_takes_the_bank_audio_in_hbm(ctx):
    """csdr_mix_push_bank: three demodulators' batch audio (mono NBFM / AM, stereo I/Q) appended to the rings by one kernel with the per-block
    peaks the audio kernel produced; the rendered mix equals the reference callback fed the fetched per-block audio and peaks."""
    import oracle.ref_modems as RM
    from cubicsdr_amd.engine import AudioMixer, DemodBank, SDRPost
    _need(RM.audio_available(), "libref_audio.so")
    fs, M, block, center, NB = 2400000, 4, 40000, 100000000, 3
    kinds = [("NBFM", 12500), ("AM", 6000), ("I/Q", 48000)]
    freqs = demod_frequencies(center, fs, len(kinds))
    post = SDRPost(ctx, fs, M, block, max_blocks=NB)
    bank = DemodBank(ctx, len(kinds), max_blocks=NB)
    for i, ((k, bw), f) in enumerate(zip(kinds, freqs)):
        bank.configure(i, post, k, bw, f)
    ref = RM.RefAudioMixCpp(48000, 3, 0)
    mix = AudioMixer(ctx, 3, 48000, ring_floats=1 << 16)
    for i, g in enumerate([1.0, 2.0, 0.5]):
        ref.set_source(i, True, g); mix.set_source(i, True, True, g, 0)
    n_cb = 0
    for rnd in range(4):
        x = synth_iq(NB * block, fs, center, [("NBFM", freqs[0]), ("AM", freqs[1]), ("NBFM", freqs[2])], seed=90 + rnd, t0=rnd * NB * block)
        post.execute(x, NB, block, center)
        bank.execute(post)
        mix.push_bank(bank, [0, 1, 2])
        for i, (k, bw) in enumerate(kinds):
            audio = bank.audio(i)
            for r in bank.results(i):
                ref.push(i, audio[r.audio_offset:r.audio_offset + r.n_audio], 2 if k == "I/Q" else 1, 48000, r.audio_peak)
        for _ in range(4):                                                # 4 x 512 frames ~ the 3 x 800 samples just produced
            want = ref.callback(512)
            got = mix.render(512, 1)
            assert np.array_equal(got, want), (rnd, float(np.max(np.abs(got - want))))
            n_cb += 1
    print("bank audio mixed in HBM: %d callbacks identical" % n_cb)
    mix.close(); ref.close(); bank.close(); post.close()


# ----------------------------------------------------------------------------------------------- PCM16 / WAV
def _wav_payload(path):
    b = open(path, "rb").read()
    assert b[:4] == b"RIFF" and b[8:16] == b"WAVEfmt " and b[36:40] == b"data"
    n = struct.unpack("<I", b[40:44])[0]
    assert n == len(b) - 44 and struct.unpack("<I", b[4:8])[0] == len(b) - 8
    return np.frombuffer(b[44:], dtype="<i2")


def test_pcm16_equals_the_reference_wav_payload(ctx):
    """csdr_bank_fetch_pcm16 (every block scaled by its own peak, on the device) and the mixer's PCM fetch against the bytes the reference's
    AudioFileWAV writes for the same AudioThreadInputs; one demodulator is driven loud enough for the anti-clipping branch (peak >= 1)."""
    import oracle.ref_modems as RM
    from cubicsdr_amd.engine import AudioMixer, DemodBank, SDRPost
    _need(RM.audio_available(), "libref_audio.so")
    fs, M, block, center, NB = 2400000, 4, 40000, 100000000, 3
    kinds = [("NBFM", 12500), ("AM", 6000), ("I/Q", 48000)]
    freqs = demod_frequencies(center, fs, len(kinds))
    post = SDRPost(ctx, fs, M, block, max_blocks=NB)
    bank = DemodBank(ctx, len(kinds), max_blocks=NB)
    for i, ((k, bw), f) in enumerate(zip(kinds, freqs)):
        bank.configure(i, post, k, bw, f)
    x = synth_iq(NB * block, fs, center, [("NBFM", freqs[0]), ("AM", freqs[1]), ("NBFM", freqs[2])], seed=123) * np.float32(6.0)   # I/Q audio peaks above 1
    post.execute(x, NB, block, center)
    bank.execute(post)
    with tempfile.TemporaryDirectory() as d:
        for i, (k, bw) in enumerate(kinds):
            audio, res = bank.audio(i), bank.results(i)
            w = RM.RefWavCpp(d, "slot%d" % i)
            for r in res:
                w.write(audio[r.audio_offset:r.audio_offset + r.n_audio], 2 if k == "I/Q" else 1, 48000, r.audio_peak)
            w.close()
            want = _wav_payload(os.path.join(d, "slot%d.wav" % i))
            got = bank.pcm16(i)
            assert got.size == want.size and np.array_equal(got, want), (k, int(np.max(np.abs(got.astype(np.int32) - want.astype(np.int32)))))
            if k == "I/Q":
                assert max(r.audio_peak for r in res) > 1.0                # the 32767 / peak branch ran
        # the mixer's output as PCM, one fixed peak
        mix = AudioMixer(ctx, 1, 48000, ring_floats=1 << 15)
        a = bank.audio(0)
        mix.push(0, a[:1000], 1, 48000, 0.4); mix.push(0, a[1000:2000], 1, 48000, 0.4)
        mix.render(256, 1)
        out = mix.render(256, 2)
        w = RM.RefWavCpp(d, "mix")
        w.write(out, 2, 48000, 1.25)
        w.close()
        assert np.array_equal(mix.pcm16(peak=1.25), _wav_payload(os.path.join(d, "mix.wav")))
        mix.close()
    bank.close(); post.close()


# ----------------------------------------------------------------------------------------------- ingest
def ingest_scenario(ctx, fs=2400000, M=4, block=40000, nb=2, rounds=5, F=2048):
    """page-locked slots -> one transfer -> the SAME device buffer feeds the channelizer and the spectrum; results equal the host-fed
    calls bit for bit, with and without the I/Q exchange, while three slots rotate under back-to-back batches"""
    from cubicsdr_amd.engine import Ingest, SDRPost, SpectrumProcessor
    center = 100000000
    x = synth_iq(rounds * nb * block, fs, center, [("NBFM", center + 300000.0), ("AM", center - 500000.0)], seed=77)
    post_a = SDRPost(ctx, fs, M, block, max_blocks=nb); post_b = SDRPost(ctx, fs, M, block, max_blocks=nb)
    spec_a = SpectrumProcessor(ctx, F, max_frames=nb); spec_b = SpectrumProcessor(ctx, F, max_frames=nb)
    ing = Ingest(ctx, nb * block, depth=3)
    for r in range(rounds):
        xb = x[r * nb * block:(r + 1) * nb * block]
        swap = r % 2 == 1
        slot = ing.acquire()
        if swap:                                             # the device delivers Q, I: the swap on the way restores I, Q
            slot[:xb.size] = (xb.imag + 1j * xb.real).astype(np.complex64)
        else:
            slot[:xb.size] = xb
        dev = ing.commit(xb.size, iq_swap=swap)
        post_a.execute(dev, nb, block, center)
        spec_a.process(dev, nb, block)
        post_b.execute(xb, nb, block, center)
        spec_b.process(xb, nb, block)
        if r >= rounds - 2:                                  # (earlier rounds are left in flight: the slot rotation is exercised without host waits)
            for ch in range(M):
                assert np.array_equal(post_a.read_channel(ch), post_b.read_channel(ch)), (r, ch)
            for k in range(nb):
                assert np.array_equal(spec_a.fetch(k)[0], spec_b.fetch(k)[0]), (r, k)
    ing.close(); spec_a.close(); spec_b.close(); post_a.close(); post_b.close()


def test_ingest_one_transfer_feeds_channelizer_and_spectrum(ctx):
    ingest_scenario(ctx)
