import sys, os, json, io, contextlib
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--cpu-seconds", "0", "--one-stream"]
import cubicsdr_amd.hip as H
import cubicsdr_amd.build as B
B.build = lambda *a, **k: None
H.LIB_PATH = sys.argv_lib = os.environ["EXPLIB"]
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
k = d["roofline"]["kernels_ms_per_step"]
print(os.environ["EXPLIB"], "frontend us:", round(k.get("demod_frontend", 0) * 1000, 1), "step ms", round(d["ms_per_step"], 4))
