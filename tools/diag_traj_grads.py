"""Diagnostic: bf16 vs fp32 gradients at each state of an fp32 training trajectory (cfg#1 shape + DINO/depth): per-step cosine, worst leaves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, spa3d
from util import O, batch_to, product_model, rel_err
cfg = O.Config(num_output_frames=24, use_dino=True, use_depth=True, dino_feature_dim=768, depth_feature_dim=1)
B, N, Q, T = 2, 64, 16, 24
batch = O.synthetic_batch(B, N, Q, T, seed=1234, dino_dim=768, depth_dim=1)
gb = batch_to(batch, 'cuda')
noise = torch.rand(B, cfg.num_latent_tokens, cfg.latent_token_dim, generator=torch.Generator().manual_seed(0)).cuda()
b32 = dict(gb); b16 = dict(gb)
for k in ('dino_features', 'depth_features'):
  b16[k] = gb[k].bfloat16(); b32[k] = gb[k].bfloat16().float()
m32 = product_model(spa3d, cfg, 'fp32'); m16 = product_model(spa3d, cfg, os.environ.get('PREC', 'bf16'))
if os.environ.get('PREC') == 'fp16':
  for k in ('dino_features', 'depth_features'): b16[k] = gb[k].bfloat16().half()
st = spa3d.TrainState(m32, m32.init(0, gb)['params'], learning_rate=float(os.environ.get('LR', 3e-4)), warmup_steps=5, total_steps=60)
for step in range(30):
  l32, g32, _ = m32.loss_and_grads({'params': st.params}, b32, noise=noise)
  _, g16, _ = m16.loss_and_grads({'params': st.params}, b16, noise=noise)
  a, b_ = g16.flat.double(), g32.flat.double()
  cos = float((a @ b_) / (a.norm() * b_.norm()))
  f16, f32 = O.tree_flatten(g16), O.tree_flatten(g32)
  gn = float(b_.norm())
  rows = []
  for k in f32:
    n = float(f32[k].double().norm())
    rows.append((float((f16[k].double() - f32[k].double()).norm()) / gn, n / gn, k))  # error as a share of the GLOBAL norm, leaf's share
  rows.sort(reverse=True)
  sig = [(e / s, s, k) for e, s, k in rows if s > 0.01]
  sig.sort(reverse=True)
  print(f'step {step:2d} loss {float(l32["total_loss"]):10.2f} |g| {gn:9.3e} cos {cos:.6f} rel {float((a-b_).norm()/b_.norm()):.4f} | top err-share: ' + ', '.join(f'{k.split("/")[0][:6]}/{"/".join(k.split("/")[1:])[-28:]}={e:.3f}(share {s:.3f})' for e, s, k in rows[:3])
        + ' | worst significant leaf rel: ' + (f'{sig[0][2][-40:]}={sig[0][0]:.3f}' if sig else '-'))
  st.train_step(b32, noise=noise)
