"""Socket power and shader clock (rocm-smi, read-only) while one kernel class runs back to back:  python tools/power_probe.py
Evidence for DESIGN.md section 5: the MFMA-dense GEMMs run the chip into its power envelope (the clock gives way), the attention / LayerNorm kernels do not."""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def sample():
  out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
  p = re.search(r'Power \(W\): ([0-9.]+)', out); c = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', out)
  return (float(p.group(1)) if p else None, float(c.group(1)) if c else None)

def probe(label, cmd):
  proc = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
  time.sleep(9)   # imports + allocation + warm-up
  ps, cs = [], []
  while proc.poll() is None and len(ps) < 40:
    p, c = sample()
    if p is not None and c is not None: ps.append(p); cs.append(c)
    else: time.sleep(0.2)
  proc.wait()
  if ps: print(f'{label:44s} samples {len(ps):3d}  power avg {sum(ps) / len(ps):7.1f} W  max {max(ps):7.1f} W   sclk avg {sum(cs) / len(cs):6.0f} MHz  min {min(cs):6.0f} MHz', flush=True)
  else: print(f'{label}: rocm-smi gave no readings', flush=True)

print(subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showmaxpower'], capture_output=True, text=True).stdout, flush=True)
probe('idle (sleep)', ['sleep', '14'])
for what, label in (('tnb', 'large-tile dW GEMM, d = 1280 shape'), ('ntb', 'large-tile NT GEMM, N = 1280 K = 2304'), ('nt8', '8-wave NT GEMM, N = 1280 K = 2304'), ('attn', 'attention backward S = 151'), ('ln', 'LayerNorm backward d = 384')):
  probe(label, [sys.executable, os.path.join(ROOT, 'tools', 'power_loop.py'), what])
