// Host-side dry runs of the C-ABI under AddressSanitizer + UndefinedBehaviorSanitizer ("ASAN on the host shim"; sanitizers on the CPU
// build only).  Built by tests/test_host_sanitizers.py: csrc/model.hip and csrc/ops.hip (the host orchestration: leaf tree, bump arena, the dry run of
// the whole forward / backward that sizes the workspace) are compiled with -fsanitize=address,undefined -fno-gpu-sanitize and linked with this
// driver and the regular objects of the kernels.  No GPU is touched: spa3d_workspace_bytes runs the orchestration with launches disabled.
// Covers the five BASELINE.json shapes, the 2-D twin, every option, and the refused configurations.
#include <cstdio>
#include <cstring>
#include <vector>

#include "spa3d.h"

static spa3d_config base(int T, int dino, int depth, int precision, int kind) {
  spa3d_config c;
  memset(&c, 0, sizeof c);
  c.num_output_frames = T; c.num_latent_tokens = 128; c.latent_token_dim = 96; c.num_frequencies = 32; c.track_scale_factor = 1.f;
  c.time_scale_factor = 150.f; c.track_token_dim = kind ? 256 : 384; c.encoder_latent_dim = 512; c.decoder_num_channels = kind ? 1024 : 1280;
  c.dino_feature_dim = dino; c.depth_feature_dim = depth; c.num_heads = 8; c.qkv_size = kind ? 512 : 768; c.enc_mlp = kind ? 1024 : 1536;
  c.enc_layers = kind ? 2 : 3; c.t2l_mlp = 2048; c.t2l_layers = kind ? 3 : 4; c.dec_mlp = 2048; c.dec_layers = kind ? 3 : 4;
  c.ro_mlp = kind ? 1024 : 1536; c.ro_layers = 4; c.precision = precision; c.model_kind = kind;
  return c;
}

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #x, __LINE__); return 1; } } while (0)

int main() {
  struct Shape { const char* name; int B, N, Q, T, dino, depth, prec, kind; };
  const Shape shapes[] = {
      {"cfg#1 B=2 64+16 T=24 xyz fp32", 2, 64, 16, 24, 0, 0, SPA3D_F32, 0},
      {"cfg#2 B=64 2048+512 T=150 C=4 bf16", 64, 2048, 512, 150, 0, 1, SPA3D_BF16, 0},
      {"cfg#3 B=64 2048+512 T=150 C=772 bf16", 64, 2048, 512, 150, 768, 1, SPA3D_BF16, 0},
      {"cfg#4 per-GPU share of B=512 (= cfg#3)", 64, 2048, 512, 150, 768, 1, SPA3D_BF16, 0},
      {"cfg#5 B=8 8192+2048 T=300 C=772 fp16", 8, 8192, 2048, 300, 768, 1, SPA3D_F16, 0},
      {"depth 256 features (inference.py:398-447)", 2, 128, 32, 150, 768, 256, SPA3D_BF16, 0},
      {"2-D TRAJAN twin", 4, 256, 64, 150, 0, 0, SPA3D_BF16, 1},
  };
  for (const Shape& s : shapes) {
    spa3d_config c = base(s.T, s.dino, s.depth, s.prec, s.kind);
    spa3d_handle h = nullptr;
    CHECK(spa3d_create(&c, &h) == SPA3D_OK && h);
    const int32_t nl = spa3d_num_leaves(h);
    const int64_t np = spa3d_param_elems(h);
    CHECK(nl > 50 && np > 1000000);
    int64_t end = 0;
    for (int32_t i = 0; i < nl; ++i) {
      char name[160]; int32_t nd = 0; int64_t shape[4] = {0, 0, 0, 0}, off = -1;
      CHECK(spa3d_leaf_info(h, i, name, &nd, shape, &off) == SPA3D_OK);
      CHECK(nd >= 1 && nd <= 4 && off >= end && strlen(name) > 0);
      int64_t n = 1; for (int k = 0; k < nd; ++k) n *= shape[k];
      end = off + n;
    }
    CHECK(end <= np);
    CHECK(spa3d_leaf_info(h, nl, nullptr, nullptr, nullptr, nullptr) == SPA3D_ERR_ARG);
    CHECK(spa3d_leaf_info(h, -1, nullptr, nullptr, nullptr, nullptr) == SPA3D_ERR_ARG);
    // workspace sizing = a dry run of the whole orchestration (forward only, then forward + backward), one sample and the whole batch at a time
    const int64_t w_f1 = spa3d_workspace_bytes(h, s.B, s.N, s.Q, s.T, 1, 0), w_t1 = spa3d_workspace_bytes(h, s.B, s.N, s.Q, s.T, 1, 1);
    const int64_t w_tb = spa3d_workspace_bytes(h, s.B, s.N, s.Q, s.T, s.B < 4 ? s.B : 4, 1);
    CHECK(w_f1 > 0 && w_t1 >= w_f1 && w_tb >= w_t1);
    // every switch, both directions, and the dry run again with the savings off (dense paths are the larger footprint or equal)
    const char* opts[] = {"prune", "ro_share", "chunk", "gemm_impl", "attn_impl", "poison"};
    for (const char* o : opts) { CHECK(spa3d_set_option(h, o, 1.0) == SPA3D_OK); CHECK(spa3d_set_option(h, o, 0.0) == SPA3D_OK); }
    for (int v = 0; v <= 6; ++v) CHECK(spa3d_set_option(h, "gemm_impl", v) == SPA3D_OK);
    for (int v = 0; v <= 4; ++v) CHECK(spa3d_set_option(h, "attn_impl", v) == SPA3D_OK);
    CHECK(spa3d_set_option(h, "gemm_impl", 0) == SPA3D_OK && spa3d_set_option(h, "attn_impl", 0) == SPA3D_OK);
    CHECK(spa3d_set_option(h, "no_such_option", 1.0) == SPA3D_ERR_ARG && strlen(spa3d_last_error(h)) > 0);
    CHECK((spa3d_set_option(h, "loss_scale", 1024.0) == SPA3D_OK) == (s.prec == SPA3D_F16));
    const int64_t w_dense = spa3d_workspace_bytes(h, s.B, s.N, s.Q, s.T, 1, 1);
    CHECK(w_dense > 0);
    CHECK(spa3d_set_option(h, "prune", 1.0) == SPA3D_OK && spa3d_set_option(h, "ro_share", 1.0) == SPA3D_OK);
    int64_t b4[4] = {-1, -1, -1, -1};
    CHECK(spa3d_grad_segments(h, b4) == SPA3D_OK && b4[0] == 0 && b4[1] > 0 && b4[2] > b4[1] && b4[3] == np);
    int64_t g4[4]; CHECK(spa3d_grad_events_recorded(h, g4) == SPA3D_OK && g4[0] == 0 && g4[2] == 0);
    int dummy_a, dummy_b; float st = 1.f;
    CHECK(spa3d_set_grad_events(h, &dummy_a, &dummy_b) == SPA3D_OK && spa3d_set_loss_scale_state(h, &st) == SPA3D_OK);
    CHECK(spa3d_detach(h, &dummy_b, &dummy_a, nullptr) == SPA3D_OK);           // not the registered pair: nothing happens
    CHECK(spa3d_grad_events_recorded(h, g4) == SPA3D_OK && g4[2] != 0);
    CHECK(spa3d_detach(h, &dummy_a, &dummy_b, &st) == SPA3D_OK);
    CHECK(spa3d_grad_events_recorded(h, g4) == SPA3D_OK && g4[2] == 0 && g4[3] == 0);
    double ps[4] = {1, 1, 1, 1}, pr[4];
    CHECK(spa3d_plan_stats(h, ps) == SPA3D_OK);
    for (int cls = 0; cls < 9; ++cls) CHECK(spa3d_prof_read(h, cls, pr) == SPA3D_OK && pr[0] == 0.0);
    CHECK(spa3d_prof_read(h, 9, pr) == SPA3D_ERR_ARG);
    printf("%-44s leaves %3d params %10lld workspace fwd/1 %7.2f GB  train/1 %7.2f GB  train/%d %7.2f GB\n", s.name, nl, (long long)np, w_f1 / 1e9, w_t1 / 1e9,
           s.B < 4 ? s.B : 4, w_tb / 1e9);
    CHECK(spa3d_destroy(h) == SPA3D_OK);
  }
  // refused configurations (attention.py:147-150 ValueError and this library's own limits); nothing may be allocated or leaked
  spa3d_handle h = nullptr;
  { spa3d_config c = base(150, 768, 1, SPA3D_BF16, 0); c.num_heads = 7; CHECK(spa3d_create(&c, &h) == SPA3D_ERR_ARG); }
  { spa3d_config c = base(150, 768, 1, SPA3D_BF16, 0); c.num_heads = 0; CHECK(spa3d_create(&c, &h) == SPA3D_ERR_ARG); }
  { spa3d_config c = base(150, 768, 1, 7, 0); CHECK(spa3d_create(&c, &h) == SPA3D_ERR_ARG); }
  { spa3d_config c = base(150, 768, 1, SPA3D_BF16, 1); CHECK(spa3d_create(&c, &h) == SPA3D_ERR_ARG); }   // 2-D twin takes no dino / depth
  { spa3d_config c = base(150, 0, 0, SPA3D_BF16, 2); CHECK(spa3d_create(&c, &h) == SPA3D_ERR_ARG); }
  { spa3d_config c = base(150, 0, 0, SPA3D_BF16, 0); c.num_frequencies = 65; CHECK(spa3d_create(&c, &h) == SPA3D_ERR_ARG); }
  { spa3d_config c = base(150, 0, 0, SPA3D_BF16, 0); c.track_token_dim = 4096; CHECK(spa3d_create(&c, &h) == SPA3D_ERR_ARG); }
  CHECK(spa3d_create(nullptr, &h) == SPA3D_ERR_ARG && h == nullptr);
  CHECK(spa3d_num_leaves(nullptr) == 0 && spa3d_param_elems(nullptr) == 0 && spa3d_workspace_bytes(nullptr, 1, 1, 1, 1, 1, 1) <= 0);
  CHECK(strlen(spa3d_version()) > 0);
  puts("HOST_DRYRUN_OK");
  return 0;
}
