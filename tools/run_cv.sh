python - <<'PY'
import os, subprocess, time, sys, json
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
def run(tag, env_extra, args):
    env = dict(os.environ); env.update(env_extra)
    t0 = time.time()
    p = subprocess.run([sys.executable, root + "/bench.py"] + args, env=env, capture_output=True, text=True)
    dt = time.time() - t0
    line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
    try:
        d = json.loads(line); ms = "%.2f ms/step %.1f img/s fwd %.1f us" % (d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"])
    except Exception as e:
        ms = "ERR " + p.stderr[-300:]
    probe = [l for l in p.stderr.splitlines() if "find-db probe" in l]
    print("%-34s wall %6.1f s   %s   | %s" % (tag, dt, ms, probe[-1] if probe else "-"), flush=True)
    return line
os.makedirs("/tmp/empty/db", exist_ok=True); os.makedirs("/tmp/empty/cache", exist_ok=True)
l = run("default (in-tree cache)", {}, ["--no_cpu_baseline"])
run("empty user db (miss expected)", {"MIOPEN_USER_DB_PATH": "/tmp/empty/db", "MIOPEN_CUSTOM_CACHE_DIR": "/tmp/empty/cache"}, ["--no_cpu_baseline", "--steps", "10", "--warmup", "5"])
PY
