// Stand-alone probe of the halo 3x3 kernel (focoos_amd/csrc/conv3x3_flat.hip): tile / wave-layout variants and ablations of
// ONE layer shape class (3x3 s1, C = N = 256: the RepVGG / res4 layers that dominate the RT-DETR step), every variant checked
// against a naive fp32 convolution and timed in interleaved rounds inside one process (guide 5.4 rule 24).
//   build: scripts/probes/build_probes.sh      run (GPU box): scripts/probes/bin/c3_probe [rounds]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../focoos_amd/csrc/conv3x3_flat.hip"
#include "../../focoos_amd/csrc/conv3x3_kplane.hip"

// entry points of other translation units that conv3x3_flat.hip's dispatcher references (never reached by the probe)
int fx_launch_pw_kplane(const ConvArgs&, const bf16_t*, hipStream_t) { return FX_ERR_UNSUPPORTED; }
bool fx_pw_kplane_supported(int, int, int) { return false; }
int fx_launch_conv3x3_c64(const ConvArgs&, const bf16_t*, hipStream_t) { return FX_ERR_UNSUPPORTED; }
bool fx_conv3x3_c64_supported(int, int, int) { return false; }
int fx_launch_conv3x3_c32(const ConvArgs&, const bf16_t*, hipStream_t) { return FX_ERR_UNSUPPORTED; }
bool fx_conv3x3_c32_supported(int, int, int, int) { return false; }

int fx_tune(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

#define HIPCHECK(x)                                                                      \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)

static uint16_t h_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float h_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// naive reference: one thread per output element, fp32 accumulation over bf16 operands, SiLU
__global__ void ref_conv3x3(const bf16_t* x, const bf16_t* w, const float* bias, float* y, int B, int H, int W, int C, int N) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long M = (long)B * H * W;
  if (idx >= M * N) return;
  const int n = idx % N;
  const long m = idx / N;
  const int xx = m % W, yy = (m / W) % H;
  float acc = bias[n];
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    if ((unsigned)(yy + dy) >= (unsigned)H || (unsigned)(xx + dx) >= (unsigned)W) continue;
    const bf16_t* xp = x + (m + dy * W + dx) * C;
    const bf16_t* wp = w + ((long)n * 9 + t) * C;
    for (int c = 0; c < C; ++c) acc += bf16_to_f32(xp[c]) * bf16_to_f32(wp[c]);
  }
  y[idx] = acc / (1.0f + __expf(-acc));
}

__global__ void cmp_kernel(const bf16_t* got, const float* ref, long n, float* out /* [0] max |err|, [1] max |ref| */) {
  float e = 0.f, r = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float g = bf16_to_f32(got[i]), f = ref[i];
    const float d = fabsf(g - f);
    e = fmaxf(e, (d == d) ? d : 1e30f);
    r = fmaxf(r, fabsf(f));
  }
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(e));
  atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(r));
}

static unsigned long long* g_dbg = nullptr;

struct Variant {
  std::string name;
  bool check;     // results comparable with the reference
  bool stamps;    // writes s_memtime stamps (ABL & 16)
  int nwaves;
  std::function<int(C3Args&, hipStream_t)> launch;
};

template <int TN, int TM, int WN, int WM>
static Variant mk(const char* name) {   // the round-2 kernel (conv3x3_flat.hip), as shipped
  return Variant{name, true, false, WN * WM, [](C3Args& a, hipStream_t s) { return launch_c3<9, 64, TN, TM, WN, WM, FX_ACT_SILU, 0>(a, s); }};
}

template <int TN, int TM, int WN, int WM, int HLP, int ABL, int LD = 1>
static Variant mkk(const char* name) {
  return Variant{name, (ABL & 15) == 0, (ABL & 16) != 0, WN * WM, [](C3Args& a, hipStream_t s) {
                   C3KArgs k{};
                   k.dbg = g_dbg;
                   (void)hipMemsetAsync(g_dbg, 0, 16 * 16 * 8, s);
                   k.x = a.x; k.wp = a.wp; k.bias = a.bias; k.res = a.res; k.y = a.y;
                   k.H = a.H; k.W = a.W; k.C = a.C; k.N = a.N; k.ldx = a.ldx; k.ldy = a.ldy; k.ldr = a.ldr; k.M = a.M;
                   k.HW = a.HW; k.y_bstride = a.y_bstride; k.x_bytes = a.x_bytes; k.r_bytes = a.r_bytes;
                   return launch_c3k<TN, TM, WN, WM, HLP, FX_ACT_SILU, 0, ABL, LD>(k, s);
                 }};
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 5;
  const int C = 256, N = 256;
  struct Shape { int B, H, W; };
  const Shape shapes[] = {{16, 40, 40}, {32, 40, 40}, {16, 80, 80}, {32, 80, 80}, {3, 20, 20}};
  std::vector<Variant> V;
  V.push_back(mk<2, 4, 4, 1>("r02 flat 4+1w 128x256           "));
  //            TN TM WN WM HLP ABL
  V.push_back(mkk<2, 4, 4, 1, 320, 0>("kplane 4+1w 128x256             "));
  V.push_back(mkk<2, 4, 4, 1, 320, 16 + 64>("kplane + stamps                 "));
  V.push_back(mkk<2, 4, 4, 1, 320, 1>("kplane abl: no weight refill    "));
  V.push_back(mkk<2, 4, 4, 1, 320, 2>("kplane abl: no pixel reads      "));
  V.push_back(mkk<2, 4, 4, 1, 320, 3>("kplane abl: no weight, no pixel "));
  V.push_back(mkk<2, 4, 4, 1, 576, 0>("kplane 128x256, HLP 576         "));
  V.push_back(mkk<2, 4, 2, 2, 512, 0>("kplane 4+1w 256x128 (2m x 2n)   "));
  V.push_back(mkk<2, 2, 4, 1, 192, 0>("kplane 4+1w  64x256 (small form) "));
  // round 6: loader-less multi-chunk forms, one halo buffer, two workgroups per CU (two waves per SIMD)
  V.push_back(mkk<2, 4, 4, 1, 320, 0, 0>("duo 4w 128x256 HLP 320          "));
  V.push_back(mkk<2, 4, 4, 1, 320, 16 + 64, 0>("duo 128x256 + stamps            "));
  V.push_back(mkk<2, 4, 4, 1, 256, 0, 0>("duo 4w 128x256 HLP 256          "));
  V.push_back(mkk<2, 2, 4, 1, 256, 0, 0>("duo 4w  64x256 HLP 256          "));
  V.push_back(mkk<2, 2, 4, 1, 192, 0, 0>("duo 4w  64x256 HLP 192          "));
  V.push_back(mkk<2, 3, 4, 1, 288, 0, 0>("duo 4w  96x256 HLP 288          "));
  V.push_back(mkk<2, 4, 4, 1, 320, 32, 0>("duo 128x256 prio by hw slot     "));
  V.push_back(mkk<2, 4, 4, 1, 320, 128, 0>("duo 128x256 prio by block parity"));
  V.push_back(mkk<2, 3, 4, 1, 288, 32, 0>("duo  96x256 prio by hw slot     "));
  V.push_back(mkk<2, 2, 4, 1, 256, 32, 0>("duo  64x256 prio by hw slot     "));

  hipStream_t st;
  HIPCHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIPCHECK(hipEventCreate(&e0));
  HIPCHECK(hipEventCreate(&e1));

  // weights [N][9][C] + fragment order [N/32][K/16][64][8], k = tap*C + c
  const int K = 9 * C;
  std::vector<uint16_t> hw((size_t)N * K), hwf((size_t)N * K);
  srand(1234);
  auto rnd = []() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; };
  for (auto& v : hw) v = h_bf16(rnd() * 0.03f);
  for (int nb = 0; nb < N / 32; ++nb)
    for (int ks = 0; ks < K / 16; ++ks)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e) hwf[(((size_t)nb * (K / 16) + ks) * 64 + l) * 8 + e] = hw[(size_t)(nb * 32 + l % 32) * K + ks * 16 + (l / 32) * 8 + e];
  std::vector<float> hb(N);
  for (auto& v : hb) v = rnd() * 0.5f;
  bf16_t *dw, *dwf;
  float* db;
  HIPCHECK(hipMalloc(&dw, hw.size() * 2));
  HIPCHECK(hipMalloc(&dwf, hwf.size() * 2));
  HIPCHECK(hipMalloc(&db, N * 4));
  HIPCHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(dwf, hwf.data(), hwf.size() * 2, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));

  for (const Shape& sh : shapes) {
    const long M = (long)sh.B * sh.H * sh.W;
    std::vector<uint16_t> hx((size_t)M * C);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = h_bf16(rnd() * 1.5f + 0.3f * ((i / C) % sh.W) / sh.W);
    bf16_t *dx, *dy;
    float *dref, *dstat;
    unsigned long long* ddbg;
    HIPCHECK(hipMalloc(&ddbg, 16 * 16 * 8));
    g_dbg = ddbg;
    HIPCHECK(hipMalloc(&dx, hx.size() * 2));
    HIPCHECK(hipMalloc(&dy, (size_t)M * N * 2));
    HIPCHECK(hipMalloc(&dref, (size_t)M * N * 4));
    HIPCHECK(hipMalloc(&dstat, 8));
    HIPCHECK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    ref_conv3x3<<<(unsigned)((M * N + 255) / 256), 256, 0, st>>>(dx, dw, db, dref, sh.B, sh.H, sh.W, C, N);
    HIPCHECK(hipStreamSynchronize(st));

    C3Args a{};
    a.x = dx; a.wp = dwf; a.bias = db; a.res = nullptr; a.y = dy;
    a.H = sh.H; a.W = sh.W; a.C = C; a.N = N; a.ldx = C; a.ldy = N; a.ldr = 0; a.M = (int)M;
    a.act = FX_ACT_SILU; a.res_after = 0; a.HLp = 0; a.HW = sh.H * sh.W; a.y_bstride = 0;
    a.x_bytes = (unsigned)(hx.size() * 2); a.r_bytes = 0;

    const double flop = 2.0 * M * N * K;
    printf("\n== B=%d H=%d W=%d  M=%ld  (%.1f GFLOP) ==\n", sh.B, sh.H, sh.W, M, flop * 1e-9);
    std::vector<std::vector<float>> times(V.size());
    std::vector<int> status(V.size(), 0);
    std::vector<float> errs(V.size(), -1.f);
    for (size_t vi = 0; vi < V.size(); ++vi) {
      HIPCHECK(hipMemsetAsync(dy, 0xff, (size_t)M * N * 2, st));
      C3Args aa = a;
      status[vi] = V[vi].launch(aa, st);
      hipError_t e = hipStreamSynchronize(st);
      if (e != hipSuccess) { fprintf(stderr, "variant %zu failed: %s\n", vi, hipGetErrorString(e)); status[vi] = -99; (void)hipGetLastError(); continue; }
      if (status[vi] != 0) continue;
      if (V[vi].stamps) {
        unsigned long long h[16 * 16];
        HIPCHECK(hipMemcpy(h, ddbg, sizeof(h), hipMemcpyDeviceToHost));
        printf("  stamps %s (cycles since wave start: first barrier | chunk starts | K loop end | tile in LDS | end)\n", V[vi].name.c_str());
        for (int w = 0; w < V[vi].nwaves + 1; ++w) {
          printf("    wave %d (+%lld):", w, (long long)(h[w * 16] - h[0]));
          for (int k = 1; k < 9; ++k) printf(" %7lld", (long long)(h[w * 16 + k] - h[w * 16]));
          printf("  | prologue: dma issued, masks, ring issued, vmcnt(0):");
          for (int k = 9; k < 13; ++k) printf(" %6lld", (long long)(h[w * 16 + k] - h[w * 16]));
          printf("\n");
        }
      }
      if (V[vi].check) {
        HIPCHECK(hipMemsetAsync(dstat, 0, 8, st));
        cmp_kernel<<<1024, 256, 0, st>>>(dy, dref, M * N, dstat);
        float hs[2];
        HIPCHECK(hipMemcpyAsync(hs, dstat, 8, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        errs[vi] = hs[0] / hs[1];
      }
    }
    const int reps = M > 150000 ? 5 : 10;
    for (int r = 0; r < rounds; ++r)
      for (size_t vi = 0; vi < V.size(); ++vi) {
        if (status[vi] != 0) continue;
        C3Args aa = a;
        V[vi].launch(aa, st);  // warm
        HIPCHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) { aa = a; V[vi].launch(aa, st); }
        HIPCHECK(hipEventRecord(e1, st));
        HIPCHECK(hipEventSynchronize(e1));
        float ms;
        HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        times[vi].push_back(ms / reps);
      }
    for (size_t vi = 0; vi < V.size(); ++vi) {
      if (status[vi] != 0) { printf("%s  status %d\n", V[vi].name.c_str(), status[vi]); continue; }
      std::sort(times[vi].begin(), times[vi].end());
      const float mn = times[vi].front(), md = times[vi][times[vi].size() / 2];
      printf("%s  min %7.1f us  med %7.1f us  %7.1f TF/s (med)  %s", V[vi].name.c_str(), mn * 1e3, md * 1e3, flop / (md * 1e-3) * 1e-12,
             V[vi].check ? "" : "[ablated]");
      if (V[vi].check) printf("relerr %.2e %s", errs[vi], errs[vi] < 1.2e-2f ? "ok" : "WRONG");
      printf("\n");
    }
    fflush(stdout);
    HIPCHECK(hipFree(dx)); HIPCHECK(hipFree(dy)); HIPCHECK(hipFree(dref)); HIPCHECK(hipFree(dstat));
  }
  return 0;
}
