import sys, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for v in sys.argv[1:]:
    env = dict(os.environ, RESEL_SSCAN_VARIANT=v)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_kernels.py'), 'sscan_only'], env=env, capture_output=True, text=True).stdout
    print('variant', v, out.strip().replace('\n', ' '))
