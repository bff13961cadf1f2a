"""Env-variable sweep of the bs=64 headline step (fresh process per setting: the library reads its diagnostics once).
    python tools/sweep64.py "CTVAE_SK_MAXWGS=0" "CTVAE_BN_BWD_MERGE_ROWS=128 CTVAE_SK_TARGET=512" ...     (on the GPU box)
Prints ms/step (best of 2 runs of 100 steps) per setting; the first line is the unmodified default."""
import json, os, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = os.environ.get("BENCH_ARGS", "").split()
for setting in [""] + sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    best = None
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-configs", "--no-roofline",
                            "--steps", "100", "--warmup", "10"] + args, env=env, capture_output=True, text=True)
        try:
            ms = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
        except Exception:
            print("FAILED", setting, r.stderr[-300:], flush=True)
            break
        best = ms if best is None else min(best, ms)
    print(f"{best}  {setting or '(default)'}", flush=True)
