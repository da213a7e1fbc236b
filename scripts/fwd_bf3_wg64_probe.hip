// VERDICT r05 item 4 asked for a BUILD of the "64 points per workgroup, LDS-resident activation tile" dataflow of the bf16x3 training
// forward, not a budget.  This probe is that dataflow's core loop on synthetic data — everything that decides its speed, nothing that
// only decides its values:
//   * one workgroup = 4 waves (one per SIMD) shares ONE tile of 64 points x 256 channels x 3 bf16 planes in LDS (3 x 64 x 528 B =
//     101 KB: the only size at which a CU holds such a tile at all), row stride padded to 528 B (conflict-free ds_read_b128);
//   * wave w owns output channels [64 w, 64 w + 64) of every layer: 2 tiles x 2 point halves = 4 accumulators (64 registers);
//   * per K-step of 16 channels: 6 A fragments (2 tiles x 3 planes, 16 B per lane) straight from global memory / L2 — no wave shares
//     them, so no LDS ring —, 6 B fragments (2 point halves x 3 planes) from the LDS tile, 24 MFMAs (the 6 cross terms i + j < 3 of
//     each of the 4 accumulators): ONE LDS read per FOUR MFMAs (the kernel in the product: one per two);
//   * layer epilogue: ReLU, three-plane split of the 64 values per lane on the VALU, ds_write_b64 of the planes into the tile for the
//     next layer — two workgroup barriers per layer (everybody has read the old tile / everybody has written the new one);
//   * STASH = 1: the fp32 activations also go out to HBM (16 B per lane per 4 channels), as the training forward must.
// Reported: achieved bf16 FLOP/s of the loop against the 2.5 PFLOP/s peak and against the product kernel's 0.457 of (peak / 6)
// fp32-equivalent.  An UPPER bound for a real kernel of this shape (no encodings, skip layer, view branch, sign bits, heads).
//   hipcc --offload-arch=gfx950 -O3 scripts/fwd_bf3_wg64_probe.hip -o scripts/fwd_bf3_wg64_probe && scripts/fwd_bf3_wg64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int W = 256, PTS = 64, NL = 8, KS = W / 16, ROWB = W * 2 + 16;   // bytes per (plane, point) row of the tile
constexpr int PLANE = PTS * ROWB;                                           // bytes per plane
constexpr int LDS_BYTES = 3 * PLANE;

__device__ __forceinline__ unsigned cvt2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// weights: [layer][kstep][wave][tile 2][plane 3][64 lanes][16 B]
template <bool STASH, int PD>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ wts, float* __restrict__ stash, float* __restrict__ out, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n = lane & 31, hh = lane >> 5;
  // initial activations: pseudo-random bf16 planes
  for (int i = threadIdx.x; i < LDS_BYTES / 4; i += 256) {
    unsigned h = 0x9e3779b9u * (i + 1 + blockIdx.x * 7919);
    h ^= h << 13; h ^= h >> 17; h ^= h << 5;
    reinterpret_cast<unsigned*>(tile)[i] = (h & 0x807f807fu) | 0x3c003c00u;
  }
  __syncthreads();
  constexpr int IA[6] = {0, 1, 2, 0, 1, 0}, IB[6] = {2, 1, 0, 1, 0, 0};
  float sink = 0.f;
  for (int rep = 0; rep < tiles_per_wg; ++rep) {
    for (int l = 0; l < NL; ++l) {
      f32x16 acc[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][h2][r] = 0.f;
      const u32x4* wl = wts + ((size_t)l * KS * 4 + w) * 6 * 64 + lane;      // + s * 4 * 6 * 64, + (t * 3 + p) * 64
      // weight fragments run PD K-steps ahead of their MFMAs (a K-step is 24 MFMAs = 768 matrix-pipe cycles; an L2 round trip is of
      // that order), activation fragments one K-step ahead (LDS)
      u32x4 A[PD + 1][6], B[2][6];
#pragma unroll
      for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int q = 0; q < 6; ++q) A[d][q] = wl[(size_t)d * 4 * 6 * 64 + q * 64];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          B[0][h2 * 3 + p] = *reinterpret_cast<const u32x4*>(tile + p * PLANE + (32 * h2 + n) * ROWB + hh * 16);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int cur = s & 1, nxt = cur ^ 1, ca = s % (PD + 1);
        if (s + PD < KS) {
#pragma unroll
          for (int q = 0; q < 6; ++q) A[(s + PD) % (PD + 1)][q] = wl[(size_t)(s + PD) * 4 * 6 * 64 + q * 64];
        }
        if (s + 1 < KS) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int p = 0; p < 3; ++p)
              B[nxt][h2 * 3 + p] = *reinterpret_cast<const u32x4*>(tile + p * PLANE + (32 * h2 + n) * ROWB + (s + 1) * 32 + hh * 16);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) acc[t][h2] = mfma(A[ca][t * 3 + IA[k]], B[cur][h2 * 3 + IB[k]], acc[t][h2]);
      }
      __syncthreads();          // every wave has read the old tile
      // epilogue: ReLU, 3-plane split, write the next layer's B operand; lane (n, hh) holds of tile t / half h2 the channels
      // 64 w + 32 t + 8 g + 4 hh + (0..3), g = 0..3, of point 32 h2 + n
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float x0 = fmaxf(acc[t][h2][4 * g], 0.f) * 0.02f, x1 = fmaxf(acc[t][h2][4 * g + 1], 0.f) * 0.02f;
            float x2 = fmaxf(acc[t][h2][4 * g + 2], 0.f) * 0.02f, x3 = fmaxf(acc[t][h2][4 * g + 3], 0.f) * 0.02f;
            if (STASH) {
              // tile-major like the product's stash: the 64 lanes of a wave write one contiguous 1 KiB block per (t, h2, g)
              float* sp = stash + ((size_t)(blockIdx.x * tiles_per_wg + rep) * NL + l) * PTS * W + (size_t)((w * 16 + (t * 2 + h2) * 4 + g) * 256) + lane * 4;
              __builtin_nontemporal_store(f32x4{x0, x1, x2, x3}, reinterpret_cast<f32x4*>(sp));
            }
            unsigned char* dst = tile + (32 * h2 + n) * ROWB + (64 * w + 32 * t + 8 * g + 4 * hh) * 2;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              const unsigned h01 = cvt2(x0, x1), h23 = cvt2(x2, x3);
              *reinterpret_cast<u32x2*>(dst + p * PLANE) = u32x2{h01, h23};
              x0 -= __uint_as_float(h01 << 16); x1 -= __uint_as_float(h01 & 0xffff0000u);
              x2 -= __uint_as_float(h23 << 16); x3 -= __uint_as_float(h23 & 0xffff0000u);
            }
            sink += x0;
          }
      __syncthreads();          // the new tile is complete
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = sink + reinterpret_cast<float*>(tile)[threadIdx.x];
}

template <bool STASH, int PD>
void run(const u32x4* wts, float* stash, float* out, int wgs, int tiles_per_wg) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<STASH, PD>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<STASH, PD>), dim3(wgs), dim3(256), LDS_BYTES, 0, wts, stash, out, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<STASH, PD>), dim3(wgs), dim3(256), LDS_BYTES, 0, wts, stash, out, tiles_per_wg);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  const double mac = (double)wgs * tiles_per_wg * NL * PTS * (double)W * W;        // fp32-equivalent MACs
  const double bf = mac * 6 * 2;
  printf("wg64 dataflow (weights %d K-steps ahead)%s  %d workgroups x %d tiles: %.3f ms  %.0f TFLOP/s bf16 = %.3f of 2500;  fp32-equivalent %.1f TFLOP/s = %.3f of 416.7 "
         "(product kernel mlp_fwd_train_bf3: 0.457)  [%s]\n", PD, STASH ? " + stash stores" : "", wgs, tiles_per_wg, ms, bf / ms / 1e9,
         bf / ms / 1e9 / 2500.0, mac * 2 / ms / 1e9, mac * 2 / ms / 1e9 / 416.7, hipGetErrorString(e));
}

int main() {
  const size_t wbytes = (size_t)NL * KS * 4 * 6 * 64 * 16;
  u32x4* wts; hipMalloc(&wts, wbytes);
  unsigned* hw = (unsigned*)malloc(wbytes);
  unsigned h = 12345u;
  for (size_t i = 0; i < wbytes / 4; ++i) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; hw[i] = (h & 0x807f807fu) | 0x3c003c00u; }
  hipMemcpy(wts, hw, wbytes, hipMemcpyHostToDevice);
  const int wgs = 256, tiles = 48;           // 256 x 48 x 64 = 786 432 points: the fine level of the C2 step
  float *stash, *out;
  hipMalloc(&stash, (size_t)wgs * tiles * NL * PTS * W * 4);
  hipMalloc(&out, (size_t)wgs * 256 * 4);
  for (int r = 0; r < 2; ++r) {
    run<false, 1>(wts, stash, out, wgs, tiles);
    run<false, 2>(wts, stash, out, wgs, tiles);
    run<false, 3>(wts, stash, out, wgs, tiles);
    run<true, 1>(wts, stash, out, wgs, tiles);
    run<true, 3>(wts, stash, out, wgs, tiles);
  }
  run<false, 3>(wts, stash, out, 512, 24);      // two workgroups per CU cannot co-reside (101 KB each): same rate expected
  return 0;
}
