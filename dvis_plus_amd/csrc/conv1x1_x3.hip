// conv1x1_x3.hip — the compute-bound 1x1 convolutions of the R50 bottlenecks (conv1 / conv3 / the stride-2 shortcuts of
// res3 - res5: detectron2 BottleneckBlock with STRIDE_IN_1X1 False) on the F16 matrix cores with split fp32
// operands: the arithmetic and the skeleton of csrc/gemm_x3.hip (fp32 operand = two f16 terms, three products per pair, fp32
// accumulation; everything transposed: out^T (co, pixel) = W (co, ci) x (ci, pixel)), on NCHW as it lies in memory:
//   * a wave owns 32 consecutive output pixels (lane & 31 = pixel, in (n, y, x) order — a tile may straddle two images) and a
//     pass of 32 NB output channels; the MFMA "B" fragment of a k-step is 8 channels of the lane's pixel: 8 dword buffer loads,
//     each 32 consecutive pixels of one channel row per half-wave (a full 128-byte line), split in registers;
//   * the input channels stream in chunks of 64 (32 loads): the chunk after the one being multiplied is in flight (the ring's
//     counted wait leaves those 32 loads out), so the activations never wait behind the weights and vice versa;
//   * weights: the pass's packed fragments through the LDS ring (global_load_lds): at 256 output channels per pass one item =
//     one activation chunk (4 k-steps x 8 blocks = 64 KB, two stages, one barrier per 96 products of a wave), otherwise
//     2 k-steps x NB blocks in three stages;
//   * work items (pixel tile, channel pass) are dealt so that the passes of one pixel tile run at the same time on ONE XCD:
//     the tile's activations come from HBM once, the other passes hit that XCD's L2;
//   * epilogue in place: + folded-BN shift, + shortcut, ReLU; a store instruction writes 32 consecutive pixels of a channel
//     per half-wave.  stride 2 (the down-sampling shortcut): the lane's pixel offset is that of input pixel (2 oy, 2 ox).
// One summation order per output: bit-reproducible, independent of the batch.
#include "dvis_common.h"
#include "x3_common.h"

namespace {

constexpr unsigned kOOB = 0x80000000u;     // beyond any served tensor (< 2 GiB): buffer loads return 0, stores are dropped

struct CxArgs {
  const float *x, *bias, *res;
  const void *wp;
  float *y;
  int N, C, K, relu, npass;
  int stride, W_in, OW, H_in;
  int taps;                                // 1: 1x1; 9: 3x3 with padding 1 (k = tap * C + channel, tap = 3 dy + dx)
  long long HW, HW_in, pixels;             // output pixels per image, input pixels per image, output pixels in total
  long long tiles;                         // ceil(pixels / (32 NW))
  float xscale, inv;
  int *flag;                               // range guard (x3_common.h)
  int tag;
  // second source of a 1x1 launch (C2 > 0): the contraction runs over [C channels of x | C2 channels of x2], x2 sampled with its
  // own stride — conv3(a) + shortcut(x) of a down-sampling bottleneck as ONE accumulation, the shortcut map never written
  const float *x2;
  int C2, stride2, W2_in;
  long long HW2_in;
  // K = 64, stride 1: the output as an OPERAND IMAGE of csrc/bneck_x3.hip instead of an fp32 map (y == NULL): per 32-pixel group
  // (n, oy, ox / 32) 8 KB = [k-step][hi, lo][32 g + ox % 32][8 halves], the output channels in accumulator order, split with oscale
  void *img;
  int XG;
  float oscale;
};

// the w-th work item of workgroup b: XCD x = b % 8 owns the tiles t = x (mod 8); its workgroups deal (tile, pass) pairs
// pass-fastest, so a tile's passes run concurrently on that XCD
__device__ __forceinline__ bool cx_item(const CxArgs &a, long long w, long long *tile, int *pass) {
  const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3, per = gridDim.x >> 3;
  const long long m = i + (long long)per * w;
  const long long u = m / a.npass;
  *pass = (int)(m - u * a.npass);
  *tile = xcd + 8 * u;
  return *tile < a.tiles;
}

// where output pixel p reads: image base (bytes), its top-left input coordinate (before the tap is added)
struct CxGeom {
  unsigned base;      // byte offset of channel 0 of image n; kOOB for a pixel past the end
  int iy, ix;         // oy * stride - pad, ox * stride - pad
};
__device__ __forceinline__ CxGeom cx_geom(const CxArgs &a, long long p) {
  CxGeom gm = {kOOB, 0, 0};
  if (p >= a.pixels) return gm;
  const long long n = p / a.HW;
  const int pix = (int)(p - n * a.HW);
  const int oy = pix / a.OW, ox = pix - oy * a.OW, pad = a.taps == 9 ? 1 : 0;
  gm.base = (unsigned)(n * a.C * a.HW_in * 4);
  gm.iy = oy * a.stride - pad, gm.ix = ox * a.stride - pad;
  return gm;
}
// byte offset of channel 0 of the input pixel tap (dy, dx) of the geometry; kOOB outside the image (zero padding)
__device__ __forceinline__ unsigned cx_tap_offset(const CxArgs &a, const CxGeom &gm, int tap) {
  const int dy = a.taps == 9 ? tap / 3 : 0, dx = a.taps == 9 ? tap - 3 * dy : 0;
  const int iy = gm.iy + dy, ix = gm.ix + dx;
  const bool ok = gm.base != kOOB && iy >= 0 && iy < a.H_in && ix >= 0 && ix < a.W_in;
  return ok ? gm.base + (unsigned)((iy * a.W_in + ix) * 4) : kOOB;
}

// byte offset of channel 0 of source 2's input pixel for output pixel p (stride2 sampling, no padding); kOOB past the end
__device__ __forceinline__ unsigned cx_geom2(const CxArgs &a, long long p) {
  if (a.C2 == 0 || p >= a.pixels) return kOOB;
  const long long n = p / a.HW;
  const int pix = (int)(p - n * a.HW);
  const int oy = pix / a.OW, ox = pix - oy * a.OW;
  return (unsigned)(n * a.C2 * a.HW2_in * 4) + (unsigned)((oy * a.stride2 * a.W2_in + ox * a.stride2) * 4);
}

// NW waves x 32 pixels per tile; 8 waves = two per SIMD (<= 256 registers each) cover each other's stalls.
// IK k-steps per ring item: 2 (three stages) or 4 (= one activation chunk; two stages of up to 64 KB: half the barriers).
template <int NB, int NW, int IK>
__global__ __launch_bounds__(NW * 64) void conv1x1_x3_kernel(const CxArgs a) {
  constexpr int kTile = NW * 32, PW = 2 * IK * NB / NW, STAGES = IK == 4 ? 2 : 3, IPC = 4 / IK;
  static_assert(2 * IK * NB % NW == 0, "the item's pieces must divide among the waves");
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  const int NCC = a.C / 64, NC1 = a.taps * NCC, NC = NC1 + a.C2 / 64, NI = IPC * NC;   // chunks per tap, of source 1, in all; ring items per work item
  long long wcount = 0;
  {
    long long t;
    int q;
    while (cx_item(a, wcount, &t, &q)) ++wcount;
  }
  if (wcount == 0) return;
  const __amdgpu_buffer_rsrc_t rx = dvis_make_rsrc_uniform(a.x, (unsigned)((long long)a.N * a.C * a.HW_in * 4));
  const __amdgpu_buffer_rsrc_t ry = dvis_make_rsrc_uniform(a.y, (unsigned)((long long)a.N * a.K * a.HW * 4));
  const __amdgpu_buffer_rsrc_t rr = dvis_make_rsrc_uniform(a.res ? a.res : a.y, (unsigned)((long long)a.N * a.K * a.HW * 4));
  const unsigned chan = (unsigned)(a.HW_in * 4);              // bytes between two input channels of a pixel
  const __amdgpu_buffer_rsrc_t rx2 = dvis_make_rsrc_uniform(a.C2 ? a.x2 : a.x, a.C2 ? (unsigned)((long long)a.N * a.C2 * a.HW2_in * 4) : 0u);
  const unsigned chan2 = (unsigned)(a.HW2_in * 4);

  typedef Ring<PW, 32, NW, STAGES> RingT;
  RingT ring;
  // issue cursor: where the next item to request lives
  long long iw = 0;
  int ic = 0, ipass;
  {
    long long t;
    cx_item(a, 0, &t, &ipass);
  }
  auto next_offset = [&]() -> size_t {
    const size_t o = ((size_t)ipass * NI + ic) * RingT::kItemBytes;
    if (++ic == NI) {
      ic = 0;
      long long t;
      if (!cx_item(a, ++iw, &t, &ipass)) ipass = 0;
    }
    return o;
  };
  ring.init(a.wp, lds, 1, (int)(wcount * NI), wave, lane);
#pragma unroll
  for (int i = 0; i < RingT::kAhead; ++i)
    if (ring.total > i) ring.issue_at(next_offset());

  // activation chunk = 64 channels of the lane's pixel: k-step s, element e -> channel 16 s + 8 g + e
  float raw[32];
  // (an out-of-range pixel has offset kOOB = 2^31; the served tensors are below 2^31 bytes, so kOOB + anything stays out of
  // range without a select — a per-load select makes hipcc branch around every load)
  auto load_raw = [&](unsigned pixoff, int cc, bool second = false) {     // `second` is wave-uniform
    const __amdgpu_buffer_rsrc_t r = second ? rx2 : rx;
    const unsigned ch = second ? chan2 : chan;
    const unsigned vo = pixoff + (unsigned)(8 * g) * ch;
    const unsigned so = (unsigned)(64 * cc) * ch;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        raw[8 * s + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, so + (unsigned)(16 * s + e) * ch, 0));
  };
  long long tile;
  int pass;
  cx_item(a, 0, &tile, &pass);
  CxGeom gm = cx_geom(a, tile * kTile + wave * 32 + j);
  unsigned gm2 = cx_geom2(a, tile * kTile + wave * 32 + j);
  load_raw(cx_tap_offset(a, gm, 0), 0);
  for (long long w = 0; w < wcount; ++w) {
    const long long p = tile * kTile + wave * 32 + j;
    // the next work item (its first chunk is requested while this one's last chunk is multiplied)
    long long ntile = tile;
    int npass_ = pass;
    const bool more = cx_item(a, w + 1, &ntile, &npass_);
    const CxGeom ngm = more ? cx_geom(a, ntile * kTile + wave * 32 + j) : CxGeom{kOOB, 0, 0};
    const unsigned ngm2 = more ? cx_geom2(a, ntile * kTile + wave * 32 + j) : kOOB;
    int tap = 0, cc = 0;                   // of the chunk that is requested next
    f16v acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
    const long long n = p / a.HW;
    const unsigned obase = p < a.pixels ? (unsigned)((n * a.K * a.HW + (p - n * a.HW)) * 4) : kOOB;
    const unsigned ochan = (unsigned)(a.HW * 4);
    const int co0 = pass * 32 * NB;
    for (int kc = 0; kc < NC; ++kc) {
      // the chunk's values are taken HERE (an opaque use: hipcc otherwise sinks the split below the requests that follow and,
      // with LDS-DMA in flight, waits for everything it finds outstanding there — the next chunk included)
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(raw[i]));
      h8 xh[4], xl[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f4 lo4 = {raw[8 * s], raw[8 * s + 1], raw[8 * s + 2], raw[8 * s + 3]};
        const f4 hi4 = {raw[8 * s + 4], raw[8 * s + 5], raw[8 * s + 6], raw[8 * s + 7]};
        split8(lo4, hi4, a.xscale, xh[s], xl[s]);
      }
      // ALWAYS 32 loads here (the ring's counted wait relies on it): the next chunk, the next item's first, or nothing (OOB)
      if (++cc == NCC) cc = 0, ++tap;
      {
        const bool first = kc + 1 < NC1, sec = !first && kc + 1 < NC;      // wave-uniform; the second source's chunks follow the first's
        const unsigned po = first ? cx_tap_offset(a, gm, tap) : (sec ? gm2 : cx_tap_offset(a, ngm, 0));
        load_raw(po, first ? cc : (sec ? kc + 1 - NC1 : 0), sec);
      }
#pragma unroll
      for (int part = 0; part < IPC; ++part) {
        const char *stage = ring.wait(false);
        ring.begin(ring.more() ? next_offset() : 0);
        mma_item<IK, NB, PW>(stage, lane, acc, xh + IK * part, xl + IK * part, [&](int i) { ring.piece(i); });
      }
    }
    // epilogue: lane = pixel; registers = channels co0 + 32 nb + 8 q + 4 g + i
    float chk = 0.f;                   // range guard: NaN as soon as one value is non-finite BEFORE the ReLU (which would hide a NaN)
    if constexpr (NB == 2) {
      if (a.img != nullptr) {
        const __amdgpu_buffer_rsrc_t ri = dvis_make_rsrc_uniform(a.img, (unsigned)((long long)a.N * a.H_in * a.XG * 8192));
        unsigned ibase = kOOB;
        if (p < a.pixels) {
          const int pix = (int)(p - n * a.HW), oy = pix / a.OW, ox = pix - oy * a.OW;
          ibase = (unsigned)(((n * a.H_in + oy) * a.XG + (ox >> 5)) * 8192 + g * 512 + (ox & 31) * 16);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float v[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f4 b = {0.f, 0.f, 0.f, 0.f};
            if (a.bias) b = *(const f4 *)(a.bias + 32 * nb + 8 * q + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float t = acc[nb][4 * q + i] * a.inv + b[i];
              chk = __builtin_fmaf(t, 0.f, chk);
              v[4 * q + i] = a.relu ? fmaxf(t, 0.f) : t;
            }
          }
          h8 hi[2], lo[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f4 p0 = {v[8 * u], v[8 * u + 1], v[8 * u + 2], v[8 * u + 3]}, p1 = {v[8 * u + 4], v[8 * u + 5], v[8 * u + 6], v[8 * u + 7]};
            split8(p0, p1, a.oscale, hi[u], lo[u]);
          }
          store_fragments4(ri, ibase, hi[0], 4096 * nb, lo[0], 4096 * nb + 1024, hi[1], 4096 * nb + 2048, lo[1], 4096 * nb + 3072);
        }
        if (a.flag != nullptr && chk != chk && p < a.pixels) atomicCAS(a.flag, 0, a.tag);
        tile = ntile, pass = npass_, gm = ngm, gm2 = ngm2;
        continue;
      }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float rv[16];                    // the shortcut's 16 values of the block as one batch of requests
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rv[r] = a.res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                      rr, obase + (unsigned)(co0 + 32 * nb + 8 * (r >> 2) + 4 * g + (r & 3)) * ochan, 0, 0))
                      : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = co0 + 32 * nb + 8 * q + 4 * g;
        f4 b = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) b = *(const f4 *)(a.bias + co);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned o = obase + (unsigned)(co + i) * ochan;
          float v = acc[nb][4 * q + i] * a.inv + b[i] + rv[4 * q + i];
          chk = __builtin_fmaf(v, 0.f, chk);
          if (a.relu) v = fmaxf(v, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, o, 0, 0);
        }
      }
    }
    if (a.flag != nullptr && chk != chk && p < a.pixels) atomicCAS(a.flag, 0, a.tag);
    tile = ntile, pass = npass_, gm = ngm, gm2 = ngm2;
  }
}

}  // namespace

DVIS_EXPORT int dvis_conv1x1_x3_supported(int C, int K, int64_t N, int64_t HW_in, int64_t HW_out) {
  if (C < 64 || C % 64 != 0 || C > 4096) return 0;
  if (!(K == 64 || K == 128 || (K > 0 && K % 256 == 0 && K <= 8192))) return 0;
  if (N <= 0 || HW_out <= 0 || HW_in < HW_out) return 0;
  if (N * C * HW_in * 4 >= ((int64_t)1 << 31) || N * K * HW_out * 4 >= ((int64_t)1 << 31)) return 0;   // 32-bit buffer offsets, kOOB
  return 1;
}

DVIS_EXPORT int64_t dvis_conv1x1_x3_packed_bytes(int C, int K) {
  if (!dvis_conv1x1_x3_supported(C, K, 1, 1, 1)) return -1;
  return (int64_t)(C / 16) * (K / 32) * 2 * kPiece;
}

/* weights (K x C) -> [pass][k-step][block][hi, lo][lane][8 halves]; wexp as in dvis_x3_pack */
DVIS_EXPORT int dvis_conv1x1_x3_pack(const float *w, int K, int C, int wexp, void *packed, void *stream) {
  DVIS_REQUIRE(w && packed, "dvis_conv1x1_x3_pack: null operand");
  DVIS_REQUIRE(dvis_conv1x1_x3_supported(C, K, 1, 1, 1), "dvis_conv1x1_x3_pack: (C, K) = (%d, %d) is not served", C, K);
  DVIS_REQUIRE(wexp >= -60 && wexp <= 60, "dvis_conv1x1_x3_pack: wexp = %d", wexp);
  const int NB = K == 64 ? 2 : K == 128 ? 4 : 8;
  const int64_t fragments = (int64_t)(C / 16) * (K / 32) * 64;
  hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((fragments + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (int64_t)C, K, C,
                     NB, 0, ldexpf(1.f, wexp), (_Float16 *)packed, fragments);
  return dvis_check_launch("dvis_conv1x1_x3_pack");
}

static int cx_launch(const float *x, const void *packed, const float *bias, const float *res, float *y, int N, int C, int K, int H, int W,
                     int stride, int taps, int xexp, int wexp, int relu, void *stream, const float *x2 = nullptr, int C2 = 0, int H2 = 0,
                     int W2 = 0, int stride2 = 1, void *image = nullptr, int oexp = 0);

DVIS_EXPORT int dvis_conv1x1_x3_dual(const float *x, const float *x2, const void *packed, const float *bias, const float *res, float *y,
                                     int N, int C, int C2, int K, int H, int W, int H2, int W2, int stride2, int xexp, int wexp, int relu,
                                     void *stream) {
  DVIS_REQUIRE(x2 != nullptr && C2 >= 64 && C2 % 64 == 0 && C2 <= 4096, "dvis_conv1x1_x3_dual: C2 %% 64 == 0 (got %d)", C2);
  DVIS_REQUIRE((stride2 == 1 || stride2 == 2) && (H2 + stride2 - 1) / stride2 == H && (W2 + stride2 - 1) / stride2 == W,
               "dvis_conv1x1_x3_dual: x2 (%d x %d, stride %d) must down-sample onto x's %d x %d map", H2, W2, stride2, H, W);
  DVIS_REQUIRE((long long)N * C2 * H2 * W2 * 4 < ((long long)1 << 31) && (uintptr_t)x2 % 16 == 0, "dvis_conv1x1_x3_dual: x2 below 2 GiB, 16-byte aligned");
  return cx_launch(x, packed, bias, res, y, N, C, K, H, W, 1, 1, xexp, wexp, relu, stream, x2, C2, H2, W2, stride2);
}

DVIS_EXPORT int dvis_conv1x1_x3(const float *x, const void *packed, const float *bias, const float *res, float *y, int N, int C, int K,
                                int H, int W, int stride, int xexp, int wexp, int relu, void *stream) {
  return cx_launch(x, packed, bias, res, y, N, C, K, H, W, stride, 1, xexp, wexp, relu, stream);
}

/* relu?(conv1x1(x, w) + bias) for K = 64 output channels, written as the operand image csrc/bneck_x3.hip reads (dvis_bneck_x3_image_bytes(N, H, W)
 * bytes; the values split with 2^oexp): the first conv1 of the res2 stage */
DVIS_EXPORT int dvis_conv1x1_x3_image(const float *x, const void *packed, const float *bias, void *image, int N, int C, int H, int W, int xexp,
                                      int wexp, int oexp, int relu, void *stream) {
  DVIS_REQUIRE(image != nullptr && (uintptr_t)image % 16 == 0, "dvis_conv1x1_x3_image: null / unaligned image");
  DVIS_REQUIRE((long long)N * H * ((W + 31) / 32) * 8192 < ((long long)1 << 31), "dvis_conv1x1_x3_image: the image must stay below 2 GiB");
  return cx_launch(x, packed, bias, nullptr, nullptr, N, C, 64, H, W, 1, 1, xexp, wexp, relu, stream, nullptr, 0, 0, 0, 1, image, oexp);
}

DVIS_EXPORT int64_t dvis_conv3x3_x3_packed_bytes(int C, int K) {
  const int64_t b = dvis_conv1x1_x3_packed_bytes(C, K);
  return b < 0 ? b : 9 * b;
}

/* weights (K x C x 3 x 3) -> [pass][k-step of (tap, channel)][block][hi, lo][lane][8 halves] */
DVIS_EXPORT int dvis_conv3x3_x3_pack(const float *w, int K, int C, int wexp, void *packed, void *stream) {
  DVIS_REQUIRE(w && packed, "dvis_conv3x3_x3_pack: null operand");
  DVIS_REQUIRE(dvis_conv1x1_x3_supported(C, K, 1, 1, 1), "dvis_conv3x3_x3_pack: (C, K) = (%d, %d) is not served", C, K);
  DVIS_REQUIRE(wexp >= -60 && wexp <= 60, "dvis_conv3x3_x3_pack: wexp = %d", wexp);
  const int NB = K == 64 ? 2 : K == 128 ? 4 : 8;
  const int64_t fragments = (int64_t)(9 * C / 16) * (K / 32) * 64;
  hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((fragments + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (int64_t)9 * C, K, 9 * C,
                     NB, 0, ldexpf(1.f, wexp), (_Float16 *)packed, fragments, 9);
  return dvis_check_launch("dvis_conv3x3_x3_pack");
}

DVIS_EXPORT int dvis_conv3x3_x3(const float *x, const void *packed, const float *bias, const float *res, float *y, int N, int C, int K,
                                int H, int W, int stride, int xexp, int wexp, int relu, void *stream) {
  return cx_launch(x, packed, bias, res, y, N, C, K, H, W, stride, 9, xexp, wexp, relu, stream);
}

static int cx_launch(const float *x, const void *packed, const float *bias, const float *res, float *y, int N, int C, int K, int H, int W,
                     int stride, int taps, int xexp, int wexp, int relu, void *stream, const float *x2, int C2, int H2, int W2, int stride2,
                     void *image, int oexp) {
  DVIS_REQUIRE(x && packed && (y || image), "dvis_conv1x1_x3: null operand");
  DVIS_REQUIRE(stride == 1 || stride == 2, "dvis_conv1x1_x3: stride %d", stride);
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
  DVIS_REQUIRE(dvis_conv1x1_x3_supported(C, K, N, (int64_t)H * W, (int64_t)OH * OW), "dvis_conv1x1_x3: shape (N %d, C %d, K %d, %d x %d, "
               "stride %d) is not served (C %% 64 == 0, K = 64 / 128 or K %% 256 == 0, tensors below 2 GiB)", N, C, K, H, W, stride);
  DVIS_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)packed % 16 == 0 && (bias == nullptr || (uintptr_t)bias % 16 == 0),
               "dvis_conv1x1_x3: operands must be 16-byte aligned");
  CxArgs a = {};
  a.x = x, a.bias = bias, a.res = res, a.wp = packed, a.y = y, a.N = N, a.C = C, a.K = K, a.relu = relu;
  a.stride = stride, a.W_in = W, a.H_in = H, a.taps = taps, a.OW = OW, a.HW = (long long)OH * OW, a.HW_in = (long long)H * W;
  a.pixels = a.HW * N;
  a.xscale = ldexpf(1.f, xexp), a.inv = ldexpf(1.f, -(xexp + wexp));
  const X3Guard gd = dvis_x3_guard();
  a.flag = gd.flag, a.tag = gd.tag;
  a.x2 = x2, a.C2 = x2 ? C2 : 0, a.stride2 = stride2, a.W2_in = W2, a.HW2_in = (long long)H2 * W2;
  a.img = image, a.XG = (OW + 31) / 32, a.oscale = ldexpf(1.f, oexp);
  const int grid = dvis_x3_persistent_cus();
  hipStream_t st = (hipStream_t)stream;
  static const int nw = getenv("DVIS_X3_CONV_WAVES") ? atoi(getenv("DVIS_X3_CONV_WAVES")) : 8;
  a.npass = K <= 128 ? 1 : K / 256;
#define DVIS_CX_LAUNCH(NBV, NWV, IKV)                                                                              \
  {                                                                                                                \
    static DvisLdsOptIn opted;                                                                                     \
    typedef Ring<2 * IKV * NBV / NWV, 32, NWV, (IKV == 4 ? 2 : 3)> R;                                              \
    a.tiles = (a.pixels + 32 * NWV - 1) / (32 * NWV);                                                              \
    const size_t lds = (IKV == 4 ? 2 : 3) * R::kItemBytes;                                                         \
    const int rc = dvis_lds_opt_in((const void *)conv1x1_x3_kernel<NBV, NWV, IKV>, lds, &opted, "dvis_conv1x1_x3"); \
    if (rc != DVIS_OK) return rc;                                                                                  \
    hipLaunchKernelGGL((conv1x1_x3_kernel<NBV, NWV, IKV>), dim3(grid), dim3(NWV * 64), lds, st, a);                \
  }
  static const int ik = getenv("DVIS_X3_CONV_ITEM") ? atoi(getenv("DVIS_X3_CONV_ITEM")) : 4;   // (2: three stages of 32 KB, 2 - 5 % slower)
  if (K == 64) {
    if (nw == 8) DVIS_CX_LAUNCH(2, 8, 2) else DVIS_CX_LAUNCH(2, 4, 2)
  } else if (K == 128) {
    if (nw == 8) DVIS_CX_LAUNCH(4, 8, 2) else DVIS_CX_LAUNCH(4, 4, 2)
  } else if (ik == 4) {
    DVIS_CX_LAUNCH(8, 8, 4)
  } else {
    if (nw == 8) DVIS_CX_LAUNCH(8, 8, 2) else DVIS_CX_LAUNCH(8, 4, 2)
  }
#undef DVIS_CX_LAUNCH
  return dvis_check_launch("dvis_conv1x1_x3");
}
