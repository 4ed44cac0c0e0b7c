"""Loss curve of the benchmarked train step itself: BASELINE configs[2] per GPU (B=64, 2048+512 tracks, T=150, C=772, bf16), N steps of TrainState on one
fixed synthetic batch (argv: steps, lr).  Evidence that the timed step trains: the loss must fall monotonically-ish from its initial value."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spa3d, bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
dev = torch.device('cuda', 0)
c = bench.CONFIGS[3]
model = spa3d.TrackAutoEncoder3D(num_output_frames=c['T'], dino_feature_dim=768, depth_feature_dim=1, precision='bf16')
batch = bench.synth_batch(c['B'], c['N'], c['Q'], c['T'], 768, 1, dev, seed=1234)
st = spa3d.TrainState(model, model.init(0, batch)['params'], learning_rate=lr, warmup_steps=5, total_steps=10 * steps)
t0 = time.perf_counter()
for i in range(steps):
  m = st.train_step(batch)
  print(f'step {i:3d} loss {float(m["train/loss"]):12.3f} position {float(m["train/position_loss"]):.6f} grad_norm {float(m["train/grad_norm"]):.4e} lr {m["train/learning_rate"]:.2e} '
        f'elapsed {time.perf_counter() - t0:7.1f} s', flush=True)
