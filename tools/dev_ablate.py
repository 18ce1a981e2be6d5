"""Developer: time the tick kernel with phases ablated (SMX_DEBUG_SKIP bits)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
masks = [0, 1, 2, 4, 8, 16, 32, 64, 1|16, 1|2|16|32|64, 255]
names = {0:'full',1:'-control',2:'-posescan',4:'-collisions',8:'-neighbours',16:'-wp sensor',32:'-corner scans',64:'-wrongway'}
E = sys.argv[1] if len(sys.argv)>1 else '1024'; N = sys.argv[2] if len(sys.argv)>2 else '8'
for m in masks:
    env = dict(os.environ, SMX_DEBUG_SKIP=str(m))
    out = subprocess.run([sys.executable, os.path.join(ROOT,'bench.py'),'--steps','40','--warmup','10','--no-cpu-baseline','--envs-per-gpu',E,'--vehicles',N], env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1]); print(f"skip={m:3d} {names.get(m,''):14s} kernel {j['roofline']['avg_kernel_ms']:.3f} ms")
    except Exception as e:
        print(m, 'failed', out.stderr[-300:])
