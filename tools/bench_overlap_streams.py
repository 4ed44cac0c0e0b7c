"""Does an MFMA-bound kernel overlap with an HBM-bound kernel on this chip, and what does a CU partition buy?

The dW GEMMs of the backward (gemm_tnb.hip) feed nothing but the optimizer: they may run on a second stream under the attention / LayerNorm backward
of the layers below.  Every persistent kernel of the library fills a CU (160 KB of LDS or all 512 registers of a lane), so two kernels only run side
by side if each is confined to a share of the CUs.  This tool measures, for A = the large-tile dW GEMM and B in {attention backward, LayerNorm backward,
attention forward, K = 384 projection GEMM (MFMA-bound control)}:
    serial           nA x A then nB x B on one stream, whole chip
    two streams      the same work on two plain streams (the hardware places workgroups as CUs free up)
    masked p/q       A on a stream confined to p of 8 CU-mask words, B on the other q (hipExtStreamCreateWithCUMask), two mask patterns
    python tools/bench_overlap_streams.py
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
lib = spa3d._lib.load()
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
torch.zeros(1, device=dev)


def hip_runtime():
  for line in open('/proc/self/maps'):
    if 'libamdhip64' in line:
      return C.CDLL(line.split()[-1])
  raise RuntimeError('libamdhip64 is not loaded')


hip = hip_runtime()
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(words):
  arr = (C.c_uint32 * len(words))(*words)
  st = C.c_void_p()
  rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
  if rc != 0 or not st.value:
    raise RuntimeError(f'hipExtStreamCreateWithCUMask -> {rc}')
  return torch.cuda.ExternalStream(st.value, device=dev)


sp = lambda st: C.c_void_p(st.cuda_stream)
ws = [torch.empty(1 << 28, dtype=torch.uint8, device=dev) for _ in range(2)]

# A: the dW GEMM of the track encoder's MLP (M = 3.07 M rows, Ki = 384, N = 1536)
M = 3065160
Xa = torch.randn(M, 384, device=dev).bfloat16(); dYa = torch.randn(M, 1536, device=dev).bfloat16()
dWa = torch.zeros(384, 1536, device=dev); Wd = torch.empty(384, 1536, device=dev, dtype=torch.bfloat16)
def op_tn(st, w): return lib.spa3d_op_linear_bwd(Xa.data_ptr(), Wd.data_ptr(), dYa.data_ptr(), None, dWa.data_ptr(), None, M, 1536, 384, 1, 0, ws[w].data_ptr(), ws[w].numel(), sp(st))

# B1: attention backward, track-encoder shape (S = 151, 8 heads of 96, qkv width 768 -> 20 298 sequences of the chunk; no key mask)
nseq, S, H, Dh = 20298, 151, 8, 96
E = H * Dh
qkv = torch.randn(nseq, S, 3 * E, device=dev).bfloat16(); sq = torch.ones(Dh, device=dev); sk = torch.ones(Dh, device=dev)
o = torch.empty(nseq, S, E, device=dev, dtype=torch.bfloat16); lse = torch.empty(nseq, H, S, 2, device=dev)
d_o = torch.randn(nseq, S, E, device=dev).bfloat16(); dqkv = torch.empty_like(qkv); dsq = torch.zeros(Dh, device=dev); dsk = torch.zeros(Dh, device=dev)
def op_attn_fwd(st, w): return lib.spa3d_op_attention(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), None, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), 1, 2, ws[w].data_ptr(), ws[w].numel(), sp(st))
def op_attn_bwd(st, w): return lib.spa3d_op_attention_bwd(qkv[..., :E].data_ptr(), qkv[..., E:2*E].data_ptr(), qkv[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), None, nseq, S, S, H, Dh, o.data_ptr(), lse.data_ptr(), d_o.data_ptr(), dqkv[..., :E].data_ptr(), dqkv[..., E:2*E].data_ptr(), dqkv[..., 2*E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), 1, 2, ws[w].data_ptr(), ws[w].numel(), sp(st))

# B2: LayerNorm backward at d = 384 over the same rows
xl = torch.randn(M, 384, device=dev).bfloat16(); scl = torch.ones(384, device=dev); stl = torch.empty(M, 2, device=dev); yl = torch.empty_like(xl)
dyl = torch.randn(M, 384, device=dev).bfloat16(); dxl = torch.empty_like(xl); dscl = torch.zeros(384, device=dev)
assert lib.spa3d_op_layernorm(xl.data_ptr(), scl.data_ptr(), yl.data_ptr(), stl.data_ptr(), M, 384, 1, sp(torch.cuda.current_stream())) == 0
def op_ln_bwd(st, w): return lib.spa3d_op_layernorm_bwd(xl.data_ptr(), scl.data_ptr(), stl.data_ptr(), dyl.data_ptr(), dxl.data_ptr(), dscl.data_ptr(), M, 384, 1, sp(st))

# B3 (control, MFMA-bound like A): the dX GEMM of the same layer, dX[M, 384] = dY[M, 1536] . W^T (NT, K = 1536)
Wn = torch.randn(384, 1536, device=dev).bfloat16(); dXn = torch.empty(M, 384, device=dev, dtype=torch.bfloat16)
def op_nt(st, w): return lib.spa3d_op_linear_bwd(Xa.data_ptr(), Wn.data_ptr(), dYa.data_ptr(), dXn.data_ptr(), None, None, M, 1536, 384, 1, 0, ws[w].data_ptr(), ws[w].numel(), sp(st))


def alone(op, st, n):
  assert op(st, 0) == 0; torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  with torch.cuda.stream(st):
    e0.record(st)
    for _ in range(n): op(st, 0)
    e1.record(st)
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n


def together(opa, na, sa, opb, nb, sb):
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); ea = torch.cuda.Event(enable_timing=True); eb = torch.cuda.Event(enable_timing=True)
  e0.record(sa); sb.wait_event(e0)
  # interleave the launches so neither queue runs dry on the host side
  ia = ib = 0
  while ia < na or ib < nb:
    if ia < na and (ib >= nb or ia * nb <= ib * na): opa(sa, 0); ia += 1
    else: opb(sb, 1); ib += 1
  ea.record(sa); eb.record(sb)
  torch.cuda.synchronize()
  return e0.elapsed_time(ea), e0.elapsed_time(eb)


def scan():
  """Throughput against the number of XCDs a kernel may use (CU-mask words 0 .. n-1): a kernel whose time does not double on half the chip is bound by something
  chip-wide (HBM, or the power envelope through the clock), not by its own instruction stream."""
  Mr = 726528
  Xr = torch.randn(Mr, 1280, device=dev).bfloat16(); dYr = torch.randn(Mr, 2304, device=dev).bfloat16(); dWr = torch.zeros(1280, 2304, device=dev)
  Wr = torch.randn(1280, 2304, device=dev).bfloat16(); dXr = torch.empty(Mr, 1280, device=dev, dtype=torch.bfloat16); Yr = torch.empty(Mr, 2304, device=dev, dtype=torch.bfloat16)
  def op_tn_r(st, w): return lib.spa3d_op_linear_bwd(Xr.data_ptr(), Wr.data_ptr(), dYr.data_ptr(), None, dWr.data_ptr(), None, Mr, 2304, 1280, 1, 0, ws[w].data_ptr(), ws[w].numel(), sp(st))
  def op_nt_r(st, w): return lib.spa3d_op_linear(Xr.data_ptr(), Wr.data_ptr(), None, None, Yr.data_ptr(), Mr, 2304, 1280, 0, 1, 0, ws[w].data_ptr(), ws[w].numel(), sp(st))
  def op_dx_r(st, w): return lib.spa3d_op_linear_bwd(Xr.data_ptr(), Wr.data_ptr(), dYr.data_ptr(), dXr.data_ptr(), None, None, Mr, 2304, 1280, 1, 0, ws[w].data_ptr(), ws[w].numel(), sp(st))
  assert op_attn_fwd(torch.cuda.current_stream(), 0) == 0
  ops = [('dW GEMM Ki=384 N=1536 (M=3.07M)', op_tn, 6), ('dW GEMM Ki=1280 N=2304 (M=727k)', op_tn_r, 6), ('NT GEMM K=1280 N=2304 (M=727k)', op_nt_r, 6),
         ('dX GEMM K=2304 N=1280 (M=727k)', op_dx_r, 6), ('dX GEMM K=1536 N=384 (M=3.07M)', op_nt, 6), ('attention backward S=151', op_attn_bwd, 2),
         ('attention forward S=151', op_attn_fwd, 4), ('LayerNorm backward d=384', op_ln_bwd, 10)]
  full = 0xFFFFFFFF
  streams = {n: masked_stream([full] * n + [0] * (8 - n)) for n in (1, 2, 4, 6, 8)}
  plain = torch.cuda.Stream(device=dev)
  print('ms per call against the number of CU-mask words (32 CUs each) the stream may use; "x" = time relative to the unmasked stream')
  for name, op, n in ops:
    t0 = alone(op, plain, n); t0 = alone(op, plain, n)
    row = [f'{name:34s} unmasked {t0:7.3f}']
    for k, st in streams.items():
      t = alone(op, st, max(1, n * k // 8)); row.append(f'{k}w {t:7.3f} (x{t / t0:.2f})')
    print('  '.join(row), flush=True)


def readout():
  """The same question at the readout stack's width (d = 1280, M = 727 k shared rows, S = 129): there the dW GEMM is bound by the power envelope, not by HBM."""
  Mr = 726528
  Xr = torch.randn(Mr, 1280, device=dev).bfloat16(); dYr = torch.randn(Mr, 2304, device=dev).bfloat16(); dWr = torch.zeros(1280, 2304, device=dev)
  Wr = torch.randn(1280, 2304, device=dev).bfloat16()
  def op_tn_r(st, w): return lib.spa3d_op_linear_bwd(Xr.data_ptr(), Wr.data_ptr(), dYr.data_ptr(), None, dWr.data_ptr(), None, Mr, 2304, 1280, 1, 0, ws[w].data_ptr(), ws[w].numel(), sp(st))
  ns, S2 = 5632, 129
  q2 = torch.randn(ns, S2, 3 * E, device=dev).bfloat16(); o2 = torch.empty(ns, S2, E, device=dev, dtype=torch.bfloat16); l2 = torch.empty(ns, H, S2, 2, device=dev)
  do2 = torch.randn(ns, S2, E, device=dev).bfloat16(); dq2 = torch.empty_like(q2)
  def fwd2(st, w): return lib.spa3d_op_attention(q2[..., :E].data_ptr(), q2[..., E:2*E].data_ptr(), q2[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), None, ns, S2, S2, H, Dh, o2.data_ptr(), l2.data_ptr(), 1, 2, ws[w].data_ptr(), ws[w].numel(), sp(st))
  def bwd2(st, w): return lib.spa3d_op_attention_bwd(q2[..., :E].data_ptr(), q2[..., E:2*E].data_ptr(), q2[..., 2*E:].data_ptr(), 3*E, 3*E, 3*E, sq.data_ptr(), sk.data_ptr(), None, ns, S2, S2, H, Dh, o2.data_ptr(), l2.data_ptr(), do2.data_ptr(), dq2[..., :E].data_ptr(), dq2[..., E:2*E].data_ptr(), dq2[..., 2*E:].data_ptr(), dsq.data_ptr(), dsk.data_ptr(), 1, 2, ws[w].data_ptr(), ws[w].numel(), sp(st))
  xr = torch.randn(Mr, 1280, device=dev).bfloat16(); scr = torch.ones(1280, device=dev); str_ = torch.empty(Mr, 2, device=dev); yr = torch.empty_like(xr)
  dyr = torch.randn(Mr, 1280, device=dev).bfloat16(); dxr = torch.empty_like(xr); dscr = torch.zeros(1280, device=dev)
  assert lib.spa3d_op_layernorm(xr.data_ptr(), scr.data_ptr(), yr.data_ptr(), str_.data_ptr(), Mr, 1280, 1, sp(torch.cuda.current_stream())) == 0
  def ln2(st, w): return lib.spa3d_op_layernorm_bwd(xr.data_ptr(), scr.data_ptr(), str_.data_ptr(), dyr.data_ptr(), dxr.data_ptr(), dscr.data_ptr(), Mr, 1280, 1, sp(st))
  plain = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
  assert fwd2(plain[0], 0) == 0
  full = 0xFFFFFFFF
  parts = {f'{a} | {8 - a} words': (masked_stream([full] * a + [0] * (8 - a)), masked_stream([0] * a + [full] * (8 - a))) for a in (2, 3, 4, 5)}
  ta = alone(op_tn_r, plain[0], 6); ta = alone(op_tn_r, plain[0], 6)
  print(f'A = dW GEMM M={Mr} Ki=1280 N=2304: {ta:.3f} ms alone on the whole chip', flush=True)
  for bname, opb in (('attention backward S=129 (5632 sequences x 8 heads)', bwd2), ('LayerNorm backward d=1280', ln2), ('attention forward S=129', fwd2)):
    tb = alone(opb, plain[1], 4); tb = alone(opb, plain[1], 4)
    na = 12; nb = max(1, round(na * ta / tb)); serial = na * ta + nb * tb
    print(f'B = {bname}: {tb:.3f} ms alone; work = {na} x A + {nb} x B = {serial:.1f} ms serial', flush=True)
    a, b = together(op_tn_r, na, plain[0], opb, nb, plain[1])
    print(f'    two plain streams:  A done {a:6.1f}  B done {b:6.1f}  makespan {max(a, b):6.1f} ms = {serial / max(a, b):.3f} x serial', flush=True)
    for name, (s0, s1) in parts.items():
      a, b = together(op_tn_r, na, s0, opb, nb, s1)
      print(f'    A | B on {name}: A done {a:6.1f}  B done {b:6.1f}  makespan {max(a, b):6.1f} ms = {serial / max(a, b):.3f} x serial', flush=True)


def main():
  if len(sys.argv) > 1 and sys.argv[1] == 'scan': return scan()
  if len(sys.argv) > 1 and sys.argv[1] == 'readout': return readout()
  plain = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
  full = 0xFFFFFFFF
  patterns = {}
  try:
    patterns['words 0-3 | 4-7'] = (masked_stream([full] * 4 + [0] * 4), masked_stream([0] * 4 + [full] * 4))
    patterns['even | odd bits'] = (masked_stream([0x55555555] * 8), masked_stream([0xAAAAAAAA] * 8))
    patterns['words 0-4 | 5-7'] = (masked_stream([full] * 5 + [0] * 3), masked_stream([0] * 5 + [full] * 3))
    patterns['words 0-2 | 3-7'] = (masked_stream([full] * 3 + [0] * 5), masked_stream([0] * 3 + [full] * 5))
  except Exception as ex:  # noqa: BLE001
    print('CU-masked streams unavailable:', ex, flush=True)
  ops_b = [('attention backward S=151', op_attn_bwd), ('LayerNorm backward d=384', op_ln_bwd), ('attention forward S=151', op_attn_fwd), ('dX GEMM K=1536 (control)', op_nt)]
  assert op_attn_fwd(plain[0], 0) == 0  # o / lse for the backward
  ta = alone(op_tn, plain[0], 8)
  print(f'A = dW GEMM M={M} Ki=384 N=1536: {ta:.3f} ms alone on the whole chip', flush=True)
  for name, (s0, s1) in patterns.items():
    print(f'    A alone on mask "{name}" first part: {alone(op_tn, s0, 4):.3f} ms   second part: {alone(op_tn, s1, 4):.3f} ms', flush=True)
  for bname, opb in ops_b:
    tb = alone(opb, plain[1], 4)
    na = 12; nb = max(1, round(na * ta / tb))
    serial = na * ta + nb * tb
    print(f'B = {bname}: {tb:.3f} ms alone; work = {na} x A + {nb} x B = {serial:.1f} ms serial', flush=True)
    a, b = together(op_tn, na, plain[0], opb, nb, plain[1])
    print(f'    two plain streams: A done {a:.1f}  B done {b:.1f}  makespan {max(a, b):.1f} ms  = {serial / max(a, b):.3f} x serial', flush=True)
    for name, (s0, s1) in patterns.items():
      tb0 = alone(opb, s1, 2)
      a, b = together(op_tn, na, s0, opb, nb, s1)
      print(f'    masks "{name}" (A | B): B alone on its part {tb0:.3f} ms; together A done {a:.1f}  B done {b:.1f}  makespan {max(a, b):.1f} ms = {serial / max(a, b):.3f} x serial', flush=True)
      a, b = together(op_tn, na, s1, opb, nb, s0)
      print(f'    masks "{name}" (B | A):                             together A done {a:.1f}  B done {b:.1f}  makespan {max(a, b):.1f} ms = {serial / max(a, b):.3f} x serial', flush=True)


main()
