import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, spa3d, bench
B, N, Q, T = 1, 8192, 2048, 300
dev = torch.device('cuda', 0)
model = spa3d.TrackAutoEncoder3D(num_output_frames=T, dino_feature_dim=768, depth_feature_dim=1, precision='bf16')
batch = bench.synth_batch(B, N, Q, T, 768, 1, dev, seed=1)
params = model.init(0, batch)['params']
st = spa3d.TrainState(model, params)
for i in range(2):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  m = st.train_step(batch)
  torch.cuda.synchronize(); print('stress step', i, time.perf_counter() - t0, 's loss', float(m['train/loss']), 'gn', float(m['train/grad_norm']), flush=True)
print('tracks/s', B * (N + Q) / (time.perf_counter() - t0))
