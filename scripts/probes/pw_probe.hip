// Stand-alone probe of the resident-tile pointwise kernel (focoos_amd/csrc/conv_pw_kplane.hip), round 6: where do the 1x1 layers of the RT-DETR
// step lose their bandwidth?  They move ~3 TB/s of algorithmic traffic where a copy moves 6.3 and the bottleneck-seam kernel 4.8; tile size,
// workgroups per CU, n-tile grouping and rotation do not change it (profiles/r06_pw_knobs.txt).  Ablations of ONE launch, interleaved rounds:
//   build: scripts/probes/build_probes.sh pw_probe.hip      run (GPU box): scripts/probes/bin/pw_probe [rounds]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../focoos_amd/csrc/conv_pw_kplane.hip"

int fx_tune(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
int fx_c3_epilogue_mode(int, bool, int) { return -1; }

#define HIPCHECK(x)                                                                      \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)

__global__ void copy_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// read two streams, write one (the traffic mix of branch2c: residual + small X in, block output out)
__global__ void rw_kernel(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = a[i];
    v.x += 1;
    b[i] = v;
  }
}

struct Variant {
  std::string name;
  std::function<int(PWKArgs&, hipStream_t)> launch;
};

template <int K, int ACT, int RM, int TM, int ABL>
static Variant mk(const char* name) {
  return Variant{name, [](PWKArgs& a, hipStream_t s) { return launch_pwk<K, ACT, RM, TM, ABL>(a, s); }};
}

template <int K, int RM>
static void run_shape(const char* title, int M, int N, int rounds, hipStream_t st) {
  // NSET copies of (x, res, y), used round-robin: every launch finds its tensors in HBM like a layer inside the step does, not in the 256 MB
  // Infinity Cache (the first version of this probe re-used ONE set: res4's branch2c read 56 us there and 78 us in the plan)
  const size_t set_bytes = (size_t)M * K * 2 + (size_t)M * N * 2 * (RM ? 2 : 1);
  const int NSET = (int)std::min<size_t>(8, std::max<size_t>(1, (size_t)900e6 / set_bytes + 1));
  bf16_t *x, *w, *res, *y;
  float* bias;
  HIPCHECK(hipMalloc(&x, (size_t)NSET * M * K * 2));
  HIPCHECK(hipMalloc(&w, (size_t)N * K * 2));
  HIPCHECK(hipMalloc(&res, (size_t)NSET * M * N * 2));
  HIPCHECK(hipMalloc(&y, (size_t)NSET * M * N * 2));
  HIPCHECK(hipMalloc(&bias, N * 4));
  HIPCHECK(hipMemset(x, 0x3c, (size_t)NSET * M * K * 2));
  HIPCHECK(hipMemset(w, 0x3c, (size_t)N * K * 2));
  HIPCHECK(hipMemset(res, 0x3c, (size_t)NSET * M * N * 2));
  HIPCHECK(hipMemset(bias, 0, N * 4));
  std::vector<Variant> V;
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 0>("as shipped (128-pixel tiles)      "));
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 1>("abl: weight ring not refilled     "));
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 2>("abl: no stores                    "));
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 4>("abl: residual not read            "));
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 8>("abl: no MFMAs                     "));
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 16>("abl: pixel tile not fetched       "));
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 1 + 8>("abl: no weights, no MFMAs (memory)"));
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 2 + 4 + 16>("abl: no X, no residual, no stores "));
  V.push_back(mk<K, FX_ACT_RELU, RM, 4, 2 + 4>("abl: no residual, no stores       "));
  V.push_back(mk<K, FX_ACT_RELU, RM, 2, 0>("64-pixel tiles                    "));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));
  const double bytes = (double)M * K * 2 + (double)M * N * 2 * (RM ? 2 : 1);
  const double flop = 2.0 * M * N * K;
  printf("\n== %s: M=%d K=%d N=%d residual=%d   %.1f MB algorithmic, %.1f GFLOP, %d buffer sets ==\n", title, M, K, N, RM, bytes * 1e-6, flop * 1e-9, NSET);
  std::vector<std::vector<float>> times(V.size() + 2);
  for (int r = 0; r < rounds; ++r) {
    for (size_t vi = 0; vi < V.size(); ++vi) {
      PWKArgs a{};
      a.x = x; a.wp = w; a.bias = bias; a.res = RM ? res : nullptr; a.y = y; a.N = N; a.ldx = K; a.ldy = N; a.ldr = N; a.M = M; a.HW = M; a.y_bstride = 0;
      a.x_bytes = (unsigned)((size_t)M * K * 2); a.r_bytes = (unsigned)std::min<size_t>((size_t)M * N * 2, 0xFFFFFFF0u);
      PWKArgs b = a;
      if (V[vi].launch(b, st) != 0) { times[vi].push_back(-1.f); continue; }
      static int rot = 0;
      HIPCHECK(hipEventRecord(e0, st));
      for (int i = 0; i < 5; ++i) {
        b = a;
        const int k = (rot++) % NSET;
        b.x = x + (size_t)k * M * K; b.res = RM ? res + (size_t)k * M * N : nullptr; b.y = y + (size_t)k * M * N;
        V[vi].launch(b, st);
      }
      HIPCHECK(hipEventRecord(e1, st));
      HIPCHECK(hipEventSynchronize(e1));
      float ms;
      HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
      times[vi].push_back(ms / 5);
    }
    // reference streams of the same byte count: a copy of the output tensor, and read-modify-write of it
    const size_t n16 = (size_t)M * N * 2 / 16;
    for (int k = 0; k < 2; ++k) {
      HIPCHECK(hipEventRecord(e0, st));
      for (int i = 0; i < 5; ++i) {
        const size_t off = (size_t)(i % NSET) * M * N * 2 / 16;
        if (k == 0) copy_kernel<<<2048, 256, 0, st>>>((const uint4*)res + off, (uint4*)y + off, n16);
        else rw_kernel<<<2048, 256, 0, st>>>((const uint4*)res + off, (uint4*)y + off, n16);
      }
      HIPCHECK(hipEventRecord(e1, st));
      HIPCHECK(hipEventSynchronize(e1));
      float ms;
      HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
      times[V.size() + k].push_back(ms / 5);
    }
  }
  for (size_t vi = 0; vi < V.size(); ++vi) {
    std::sort(times[vi].begin(), times[vi].end());
    const float md = times[vi][times[vi].size() / 2];
    if (md < 0) { printf("%s  unsupported\n", V[vi].name.c_str()); continue; }
    printf("%s  %7.1f us   %6.2f TB/s algorithmic   %7.1f TF/s\n", V[vi].name.c_str(), md * 1e3, bytes / (md * 1e-3) * 1e-12, flop / (md * 1e-3) * 1e-12);
  }
  for (int k = 0; k < 2; ++k) {
    auto& t = times[V.size() + k];
    std::sort(t.begin(), t.end());
    printf("%s  %7.1f us   %6.2f TB/s (read + write of the [M][N] tensor)\n", k == 0 ? "plain copy of the output tensor     " : "read-modify-write of it            ", t[t.size() / 2] * 1e3,
           2.0 * M * N * 2 / (t[t.size() / 2] * 1e-3) * 1e-12);
  }
  fflush(stdout);
  HIPCHECK(hipFree(x)); HIPCHECK(hipFree(w)); HIPCHECK(hipFree(res)); HIPCHECK(hipFree(y)); HIPCHECK(hipFree(bias));
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  hipStream_t st;
  HIPCHECK(hipStreamCreate(&st));
  run_shape<256, 1>("res4 branch2c (bs 32)", 51200, 1024, rounds, st);
  run_shape<256, 1>("res4 branch2c (bs 16)", 25600, 1024, rounds, st);
  run_shape<512, 1>("res5 branch2c (bs 32)", 12800, 2048, rounds, st);
  run_shape<256, 0>("value projection (bs 32)", 268800, 1536, rounds, st);
  run_shape<512, 0>("CSP conv1|conv2 at 80x80 (bs 32)", 204800, 512, rounds, st);
  return 0;
}
