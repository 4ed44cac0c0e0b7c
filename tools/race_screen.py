"""Race screen for the counted-vmcnt kernels (cdna_hip_programming.md: "screen a new sync structure for races over many runs at
several sizes").  The NT kernels have no atomics, so every run of one problem must be BIT-identical; a read that raced its LDS-DMA
shows up as a run-to-run difference (and as an error against the fp64 reference).  The TN kernels accumulate with f32 atomics: checked
against the fp64 reference with a tight tolerance.  REPS runs per shape (default 40)."""
import ctypes as C, sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d
def run(REPS=40, verbose=True):
  bad = 0
  lib = spa3d._lib.load()
  s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
  ws = torch.empty(1 << 28, dtype=torch.uint8, device='cuda')
  g = torch.Generator(device='cuda').manual_seed(7)
  nt_shapes = [(70000, 512, 256, 0, True), (66048, 768, 384, 1, False), (140000, 256, 128, 0, False), (256 * 700, 2304, 384, 0, False),
               (131072, 1280, 768, 0, True), (99991, 384, 768, 0, True), (65600, 384, 1536, 1, False), (262144, 1536, 384, 1, False),
               (80000, 1280, 1536, 0, False), (65536, 256, 64, 0, False), (100000, 384, 768, 0, True), (131072, 384, 2304, 0, False),
               (66000, 1152, 384, 1, False)]
  for (M, N, K, act, res) in nt_shapes:
    A = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    B = (torch.randn(K, N, device='cuda', generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device='cuda', generator=g)
    R = torch.randn(M, N, device='cuda', generator=g).bfloat16() if res else None
    outs = []
    first = None; ndiff = 0
    for r in range(REPS):
      Cc = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
      rc = lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), bias.data_ptr(), R.data_ptr() if res else None, Cc.data_ptr(), M, N, K, act, 1, 2,
                               ws.data_ptr(), ws.numel(), s())
      assert rc == 0
      if first is None: first = Cc
      else: ndiff += int((Cc.view(torch.int16) != first.view(torch.int16)).sum())
    # reference on a row sample (fp64 on the GPU would need M*N*8 bytes)
    idx = torch.randint(0, M, (2048,), device='cuda', generator=g)
    ref = A[idx].double() @ B.double() + bias.double()
    if act:
      ref = 0.5 * ref * (1 + torch.tanh(0.7978845608028654 * (ref + 0.044715 * ref ** 3)))
    if res: ref = ref + R[idx].double()
    err = float((first[idx].double() - ref).abs().max() / ref.abs().max())
    nanc = int(torch.isnan(first.float()).sum())
    ok = ndiff == 0 and err < 8e-3 and nanc == 0
    bad += not ok
    if verbose: print(f'NT M={M:7d} N={N:5d} K={K:5d} act={act} res={int(res)}  run-to-run differing elements {ndiff}  max rel err {err:.2e}  nan {nanc}  {"ok" if ok else "FAIL"}', flush=True)
    del A, B, R, first
  # round 4: the row-stationary K = 384 GEMM (impl 7; with the gelu'-multiply epilogue: act 2) and the fused MLP forward -- LDS-DMA rings with counted vmcnt and a staging
  # image inside the ring slot that is being refilled; no atomics, so every run must be bit-identical
  for (M, N, aux) in [(70000, 2304, False), (256 * 300 + 17, 768, False), (131072, 1536, True), (66000, 1536, False), (50001, 2304, True)]:
    A = torch.randn(M, 384, device='cuda', generator=g).bfloat16()
    B = (torch.randn(384, N, device='cuda', generator=g) / math.sqrt(384)).bfloat16()
    bias = torch.randn(N, device='cuda', generator=g)
    P = torch.randn(M, N, device='cuda', generator=g).bfloat16() if aux else None
    first = None; ndiff = 0
    for r in range(REPS):
      Cc = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
      rc = lib.spa3d_op_linear(A.data_ptr(), B.data_ptr(), None if aux else bias.data_ptr(), P.data_ptr() if aux else None, Cc.data_ptr(), M, N, 384, 2 if aux else 0, 1, 7,
                               ws.data_ptr(), ws.numel(), s())
      assert rc == 0
      if first is None: first = Cc
      else: ndiff += int((Cc.view(torch.int16) != first.view(torch.int16)).sum())
    idx = torch.randint(0, M, (2048,), device='cuda', generator=g)
    ref = A[idx].double() @ B.double()
    if aux:
      x = P[idx].double(); c = 0.7978845608028654; t = torch.tanh(c * (x + 0.044715 * x ** 3))
      ref = ref * (0.5 * (1 + t) + 0.5 * x * (1 - t * t) * c * (1 + 3 * 0.044715 * x * x))
    else: ref = ref + bias.double()
    err = float((first[idx].double() - ref).abs().max() / ref.abs().max())
    nanc = int(torch.isnan(first.float()).sum())
    ok = ndiff == 0 and err < 8e-3 and nanc == 0
    bad += not ok
    if verbose: print(f'RS M={M:7d} N={N:5d} K=  384 aux={int(aux)}  run-to-run differing elements {ndiff}  max rel err {err:.2e}  nan {nanc}  {"ok" if ok else "FAIL"}', flush=True)
    del A, B, P, first
  for M in (128 * 600 + 40, 50000):
    d, mlp = 384, 1536
    na = torch.randn(M, d, device='cuda', generator=g).bfloat16(); a = torch.randn(M, d, device='cuda', generator=g).bfloat16()
    w_in = (torch.randn(d, mlp, device='cuda', generator=g) / math.sqrt(d)).bfloat16(); w_out = (torch.randn(mlp, d, device='cuda', generator=g) / math.sqrt(mlp)).bfloat16()
    b_in = torch.randn(mlp, device='cuda', generator=g); b_out = torch.randn(d, device='cuda', generator=g)
    first = None; ndiff = 0
    for r in range(REPS):
      y = torch.full((M, d), float('nan'), device='cuda', dtype=torch.bfloat16); h = torch.full((M, mlp), float('nan'), device='cuda', dtype=torch.bfloat16); hp = torch.full_like(h, float('nan'))
      rc = lib.spa3d_op_mlp_fused(na.data_ptr(), a.data_ptr(), w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(), b_out.data_ptr(), y.data_ptr(), h.data_ptr(), hp.data_ptr(),
                                  M, d, mlp, 1, ws.data_ptr(), ws.numel(), s())
      assert rc == 0
      cur = (y, h, hp)
      if first is None: first = cur
      else: ndiff += sum(int((c.view(torch.int16) != f.view(torch.int16)).sum()) for c, f in zip(cur, first))
    nanc = sum(int(torch.isnan(t.float()).sum()) for t in first)
    ok = ndiff == 0 and nanc == 0
    bad += not ok
    if verbose: print(f'MLP fused M={M:7d}  run-to-run differing elements (y, h, hpre) {ndiff}  nan {nanc}  {"ok" if ok else "FAIL"}', flush=True)
    del na, a, first
  tn_shapes = [(70000, 768, 384), (131072, 768, 1280), (99968, 384, 1536), (65536, 1536, 1280), (262144, 384, 768), (80000, 1280, 768)]
  for (M, N, K) in tn_shapes:   # dB[K][N] = A[M][K]^T dC[M][N]
    A = torch.randn(M, K, device='cuda', generator=g).bfloat16()
    dC = torch.randn(M, N, device='cuda', generator=g).bfloat16()
    Bd = torch.empty(K, N, device='cuda', dtype=torch.bfloat16)
    ref = (A.double().T @ dC.double())
    for impl, kname in ((2, 'large register tile (round 5)'), (8, '8-wave kernels')):   # every shape here divides the 384 x 256 tile: impl 2 takes the large-tile kernel, 8 the 8-wave ones
      worst = 0.0
      for r in range(REPS):
        dB = torch.full((K, N), float('nan'), device='cuda')
        rc = lib.spa3d_op_linear_bwd(A.data_ptr(), Bd.data_ptr(), dC.data_ptr(), None, dB.data_ptr(), None, M, N, K, 1, impl, ws.data_ptr(), ws.numel(), s())
        assert rc == 0
        worst = max(worst, float((dB.double() - ref).abs().max() / ref.abs().max()))
      ok = worst < 2e-5
      bad += not ok
      if verbose: print(f'TN M={M:7d} N={N:5d} Ki={K:5d} {kname:30s} worst max rel err over {REPS} runs {worst:.2e}  {"ok" if ok else "FAIL"}', flush=True)
    del A, dC, ref
  if verbose: print('RACE SCREEN', 'FAILED' if bad else 'clean', flush=True)
  return bad


if __name__ == '__main__':
  sys.exit(1 if run(int(os.environ.get('REPS', 40))) else 0)
