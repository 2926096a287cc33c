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

DVIS_EXPORT int64_t dvis_conv_x3_image_bytes(int64_t N, int C, int H, int W);

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
  int OH;                                  // output rows
  // IMGIN: the input as an operand image (written by an IMGOUT launch, dvis_upsample_add_image or csrc/bneck_x3.hip) instead of
  // x: per 32-pixel group (n, iy, ix / 32) C / 64 chunks of 8 KB, the channels of a chunk in accumulator order — the weights
  // are packed to match (dvis_conv_x3_pack_image).  Nothing is split: a chunk is 8 loads of 16 bytes per lane.
  const void *ximg;
  int XG_in;
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
  int row0;           // n * H_in: the image's first row in an operand image
};
__device__ __forceinline__ CxGeom cx_geom(const CxArgs &a, long long p) {
  CxGeom gm = {kOOB, 0, 0, 0};
  if (p >= a.pixels) return gm;
  const long long n = p / a.HW;
  const int pix = (int)(p - n * a.HW);
  const int oy = pix / a.OW, ox = pix - oy * a.OW, pad = a.taps == 9 ? 1 : 0;
  gm.base = (unsigned)(n * a.C * a.HW_in * 4);
  gm.iy = oy * a.stride - pad, gm.ix = ox * a.stride - pad;
  gm.row0 = (int)(n * a.H_in);
  return gm;
}
// byte offset of channel 0 of the input pixel tap (dy, dx) of the geometry; kOOB outside the image (zero padding)
__device__ __forceinline__ unsigned cx_tap_offset(const CxArgs &a, const CxGeom &gm, int tap) {
  const int dy = a.taps == 9 ? tap / 3 : 0, dx = a.taps == 9 ? tap - 3 * dy : 0;
  const int iy = gm.iy + dy, ix = gm.ix + dx;
  const bool ok = gm.base != kOOB && iy >= 0 && iy < a.H_in && ix >= 0 && ix < a.W_in;
  return ok ? gm.base + (unsigned)((iy * a.W_in + ix) * 4) : kOOB;
}

// IMGIN: byte offset of the lane's 16 bytes (k-step 0, high term) in chunk 0 of the group that holds input pixel tap (dy, dx)
__device__ __forceinline__ unsigned cx_tap_offset_img(const CxArgs &a, const CxGeom &gm, int tap, int g) {
  const int dy = a.taps == 9 ? tap / 3 : 0, dx = a.taps == 9 ? tap - 3 * dy : 0;
  const int iy = gm.iy + dy, ix = gm.ix + dx;
  const bool ok = gm.base != kOOB && iy >= 0 && iy < a.H_in && ix >= 0 && ix < a.W_in;
  return ok ? (unsigned)(((gm.row0 + iy) * a.XG_in + (ix >> 5)) * ((a.C / 64) * 8192) + g * 512 + (ix & 31) * 16) : kOOB;
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
// IMGIN / IMGOUT: the input / output as operand images (CxArgs::ximg / img) — the 64 .. 512-channel maps INSIDE a bottleneck and the
// FPN output convolution's input travel that way: the producer's epilogue splits once, the nine taps of a 3x3 consumer re-read the
// image (8 loads of 16 bytes per chunk and lane instead of 32 of 4 bytes, no split arithmetic in the loop).
template <int NB, int NW, int IK, bool IMGIN = false, bool IMGOUT = false>
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

  typedef Ring<PW, IMGIN ? 8 : 32, NW, STAGES> RingT;
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
  // IMGIN: the next chunk's operand pair, requested as it will be multiplied
  const __amdgpu_buffer_rsrc_t rxi = dvis_make_rsrc_uniform(IMGIN ? a.ximg : (const void *)a.x,
                                                            IMGIN ? (unsigned)((long long)a.N * a.H_in * a.XG_in * (a.C / 64) * 8192) : 0u);
  h8 nh[4], nl[4];
  auto load_img = [&](unsigned off, int cc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      nh[s] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rxi, off, (unsigned)(8192 * cc + 2048 * s), 0));
      nl[s] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rxi, off, (unsigned)(8192 * cc + 2048 * s + 1024), 0));
    }
  };
  long long tile;
  int pass;
  cx_item(a, 0, &tile, &pass);
  CxGeom gm = cx_geom(a, tile * kTile + wave * 32 + j);
  unsigned gm2 = cx_geom2(a, tile * kTile + wave * 32 + j);
  if constexpr (IMGIN)
    load_img(cx_tap_offset_img(a, gm, 0, g), 0);
  else
    load_raw(cx_tap_offset(a, gm, 0), 0);
  for (long long w = 0; w < wcount; ++w) {
    const long long p = tile * kTile + wave * 32 + j;
    // the next work item (its first chunk is requested while this one's last chunk is multiplied)
    long long ntile = tile;
    int npass_ = pass;
    const bool more = cx_item(a, w + 1, &ntile, &npass_);
    const CxGeom ngm = more ? cx_geom(a, ntile * kTile + wave * 32 + j) : CxGeom{kOOB, 0, 0, 0};
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
      h8 xh[4], xl[4];
      if constexpr (IMGIN) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          asm volatile("" : "+v"(nh[s]), "+v"(nl[s]));
          xh[s] = nh[s], xl[s] = nl[s];
        }
        // ALWAYS 8 loads here (the ring's counted wait relies on it): the next chunk, the next item's first, or nothing (OOB)
        if (++cc == NCC) cc = 0, ++tap;
        const bool first = kc + 1 < NC1;
        load_img(first ? cx_tap_offset_img(a, gm, tap, g) : cx_tap_offset_img(a, ngm, 0, g), first ? cc : 0);
      } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(raw[i]));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const f4 lo4 = {raw[8 * s], raw[8 * s + 1], raw[8 * s + 2], raw[8 * s + 3]};
          const f4 hi4 = {raw[8 * s + 4], raw[8 * s + 5], raw[8 * s + 6], raw[8 * s + 7]};
          split8(lo4, hi4, a.xscale, xh[s], xl[s]);
        }
        // ALWAYS 32 loads here (the ring's counted wait relies on it): the next chunk, the next item's first, or nothing (OOB)
        if (++cc == NCC) cc = 0, ++tap;
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
    if constexpr (NB == 2 || IMGOUT) {
      if (a.img != nullptr) {
        // K / 64 chunks per group; this pass holds chunks co0 / 64 .. : block nb = k-steps 2 (nb & 1), 2 (nb & 1) + 1 of chunk nb / 2
        const unsigned gs = (unsigned)(a.K / 64) * 8192u;
        const __amdgpu_buffer_rsrc_t ri = dvis_make_rsrc_uniform(a.img, (unsigned)((long long)a.N * a.OH * a.XG * gs));
        const __amdgpu_buffer_rsrc_t rb = dvis_make_rsrc_uniform(a.bias ? (const void *)a.bias : (const void *)a.img, a.bias ? (unsigned)a.K * 4u : 0u);
        unsigned ibase = kOOB;
        if (p < a.pixels) {
          const int pix = (int)(p - n * a.HW), oy = pix / a.OW, ox = pix - oy * a.OW;
          ibase = (unsigned)(((n * a.OH + oy) * a.XG + (ox >> 5)) * gs + (co0 / 64) * 8192 + g * 512 + (ox & 31) * 16);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float v[16];
          // (the shifts through a descriptor that is empty without a bias: no branch per load — with 32 conditional loads at NB = 8
          // hipcc hoisted them all to the epilogue's head and spilled 60 - 100 registers around them)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f4 b = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(co0 + 32 * nb + 8 * q + 4 * g) * 4u, 0, 0));
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
          store_fragments4(ri, ibase, hi[0], 4096 * nb, lo[0], 4096 * nb + 1024, hi[1], 4096 * nb + 2048, lo[1], 4096 * nb + 3072);      // (8192 (nb / 2) + 2048 (2 (nb & 1) + u) + 1024 hl)
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

// lateral * scale + shift + bilinear(top) (the FPN's top-down sum, csrc/fused_elementwise.hip: upsample_add_kernel — the same
// formula and operation order) written as an OPERAND IMAGE of C channels: the 3x3 output convolution behind it (IMGIN) reads
// fragments instead of splitting the fp32 map nine times.  One workgroup = one (32-pixel group, 64-channel chunk): thread
// (k-step S, half g, pixel x) makes the 8 channels of its fragment; a wave's store is 1 KB of consecutive image bytes.
__global__ __launch_bounds__(256) void upsample_add_image_kernel(const float *__restrict__ lateral, const float *__restrict__ top,
                                                                 void *__restrict__ img, int N, int C, int H, int W, int h, int w,
                                                                 const float *__restrict__ lat_scale, const float *__restrict__ lat_shift,
                                                                 float oscale) {
  // grid (chunk, group of the row, block of 16 image rows): no per-thread division beyond row -> (image, y)
  const int NCC = C / 64, XG = (W + 31) / 32;
  const int cc = blockIdx.x, xg = blockIdx.y;
  const int x = threadIdx.x & 31, g = (threadIdx.x >> 5) & 1, S = threadIdx.x >> 6;
  const int px = 32 * xg + x;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const int pxc = px < W ? px : W - 1;               // (lanes past the row's end compute a value nobody reads)
  float fx = sx * ((float)pxc + 0.5f) - 0.5f;
  fx = fx < 0.f ? 0.f : fx;
  const int x0 = min((int)fx, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
  const __amdgpu_buffer_rsrc_t ri = dvis_make_rsrc_uniform(img, (unsigned)((long long)N * H * XG * NCC * 8192));
  const unsigned rows = (unsigned)N * (unsigned)H;
  for (unsigned row = blockIdx.z * 16u; row < blockIdx.z * 16u + 16u && row < rows; ++row) {
    const unsigned n = row / (unsigned)H;
    const int y = (int)(row - n * (unsigned)H);
    float fy = sy * ((float)y + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = min((int)fy, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 64 * cc + 32 * (S >> 1) + 16 * (S & 1) + 8 * (e >> 2) + 4 * g + (e & 3);
      const unsigned pl = n * (unsigned)C + (unsigned)c;
      const float *t0 = top + ((size_t)pl * h + y0) * (size_t)w, *t1 = top + ((size_t)pl * h + y1) * (size_t)w;
      float lat = lateral[((size_t)pl * H + y) * (size_t)W + pxc];
      if (lat_scale) lat = lat * lat_scale[pl] + lat_shift[pl];
      v[e] = lat + (ly0 * (lx0 * t0[x0] + lx1 * t0[x1]) + ly1 * (lx0 * t1[x0] + lx1 * t1[x1]));
    }
    const f4 p0 = {v[0], v[1], v[2], v[3]}, p1 = {v[4], v[5], v[6], v[7]};
    h8 hi, lo;
    split8(p0, p1, oscale, hi, lo);
    const unsigned G = row * (unsigned)XG + (unsigned)xg;
    const unsigned off = (G * (unsigned)NCC + (unsigned)cc) * 8192u + (unsigned)(S * 2048 + g * 512 + x * 16);
    store_fragments2(ri, off, hi, 0, lo, 1024);
  }
}

}  // namespace

/* image (C channels, dvis_conv_x3_image_bytes(N, C, H, W)) = split(lateral * scale + shift + bilinear(top)): dvis_upsample_add[_affine]
 * with the result as an operand image for an image-input convolution; lat_scale / lat_shift per plane or NULL */
DVIS_EXPORT int dvis_upsample_add_image(const float *lateral, const float *lat_scale, const float *lat_shift, const float *top, void *image,
                                        int N, int C, int H, int W, int h, int w, int oexp, void *stream) {
  DVIS_REQUIRE(lateral && top && image, "dvis_upsample_add_image: null pointer");
  DVIS_REQUIRE((lat_scale == nullptr) == (lat_shift == nullptr), "dvis_upsample_add_image: scale and shift together");
  DVIS_REQUIRE(N > 0 && C > 0 && C % 64 == 0 && H > 0 && W > 0 && h > 0 && w > 0, "dvis_upsample_add_image: bad sizes");
  DVIS_REQUIRE((uintptr_t)image % 16 == 0 && dvis_conv_x3_image_bytes(N, C, H, W) < ((int64_t)1 << 31), "dvis_upsample_add_image: a 16-byte "
               "aligned image below 2 GiB");
  const int per = 16 * 65535 / H;    // images per launch: the grid's z extent is the image rows / 16
  DVIS_REQUIRE(per >= 1 && (W + 31) / 32 <= 65535, "dvis_upsample_add_image: map too large (%d x %d)", H, W);
  for (int i = 0; i < N; i += per) {
    const int n = N - i < per ? N - i : per;
    hipLaunchKernelGGL(upsample_add_image_kernel, dim3(C / 64, (W + 31) / 32, (n * H + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                       lateral + (size_t)i * C * H * W, top + (size_t)i * C * h * w,
                       (char *)image + (size_t)i * H * ((W + 31) / 32) * (C / 64) * 8192, n, C, H, W, h, w,
                       lat_scale ? lat_scale + (size_t)i * C : nullptr, lat_shift ? lat_shift + (size_t)i * C : nullptr, ldexpf(1.f, oexp));
  }
  return dvis_check_launch("dvis_upsample_add_image");
}

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
                     int W2 = 0, int stride2 = 1, void *image = nullptr, int oexp = 0, const void *ximg = nullptr);

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

/* Operand images with C channels: per 32-pixel group (n, y, x / 32) C / 64 chunks of 8 KB (dvis_conv_x3_image_bytes). */
DVIS_EXPORT int64_t dvis_conv_x3_image_bytes(int64_t N, int C, int H, int W) {
  if (N < 0 || C <= 0 || C % 64 != 0 || H <= 0 || W <= 0) return -1;
  return N * H * ((W + 31) / 32) * (int64_t)(C / 64) * 8192;
}

/* weights (K, C, taps: 1 or 9) for an image-input launch: the channels of every 64-chunk in ACCUMULATOR order (the order in which the
 * producer's lanes hold them), otherwise the layout of dvis_conv1x1_x3_pack / dvis_conv3x3_x3_pack (same packed size) */
DVIS_EXPORT int dvis_conv_x3_pack_image(const float *w, int K, int C, int taps, int wexp, void *packed, void *stream) {
  DVIS_REQUIRE(w && packed, "dvis_conv_x3_pack_image: null operand");
  DVIS_REQUIRE(taps == 1 || taps == 9, "dvis_conv_x3_pack_image: taps = %d", taps);
  DVIS_REQUIRE(dvis_conv1x1_x3_supported(C, K, 1, 1, 1) && K >= 128, "dvis_conv_x3_pack_image: (C, K) = (%d, %d) is not served", C, K);
  DVIS_REQUIRE(wexp >= -60 && wexp <= 60, "dvis_conv_x3_pack_image: wexp = %d", wexp);
  const int NB = K == 128 ? 4 : 8;
  const int64_t fragments = (int64_t)(taps * C / 16) * (K / 32) * 64;
  hipLaunchKernelGGL(x3_pack_kernel, dim3((unsigned)((fragments + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (int64_t)taps * C, K,
                     taps * C, NB, 1, ldexpf(1.f, wexp), (_Float16 *)packed, fragments, taps);
  return dvis_check_launch("dvis_conv_x3_pack_image");
}

/* relu?(conv(x) + bias + res) with the INPUT and / or the OUTPUT as operand images (csrc/conv1x1_x3.hip, IMGIN / IMGOUT):
 *   ximg != NULL: input image of C channels on the H x W map (x ignored), weights from dvis_conv_x3_pack_image; else x (N, C, H, W) and
 *                 weights from dvis_conv1x1_x3_pack / dvis_conv3x3_x3_pack;
 *   image != NULL: output image of K channels (y, res ignored), the values split with 2^oexp; else y (N, K, OH, OW).
 * xexp: the exponent the INPUT was / is split with.  K = 128 or K %% 256 == 0; taps 1 (1x1) or 9 (3x3, padding 1); stride 1 or 2. */
DVIS_EXPORT int dvis_conv_x3_image(const void *ximg, const float *x, const void *packed, const float *bias, const float *res, float *y,
                                   void *image, int N, int C, int K, int H, int W, int stride, int taps, int xexp, int wexp, int oexp, int relu,
                                   void *stream) {
  DVIS_REQUIRE(ximg != nullptr || image != nullptr, "dvis_conv_x3_image: neither side is an operand image (use dvis_conv1x1_x3 / dvis_conv3x3_x3)");
  DVIS_REQUIRE((ximg != nullptr) != (x != nullptr), "dvis_conv_x3_image: exactly one of ximg and x");
  DVIS_REQUIRE((image != nullptr) != (y != nullptr), "dvis_conv_x3_image: exactly one of image and y");
  DVIS_REQUIRE(K == 128 || K % 256 == 0, "dvis_conv_x3_image: K = %d (128 or a multiple of 256)", K);
  DVIS_REQUIRE(taps == 1 || taps == 9, "dvis_conv_x3_image: taps = %d", taps);
  DVIS_REQUIRE(image == nullptr || res == nullptr, "dvis_conv_x3_image: an image output takes no residual");
  DVIS_REQUIRE(((uintptr_t)ximg | (uintptr_t)image) % 16 == 0, "dvis_conv_x3_image: images must be 16-byte aligned");
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
  DVIS_REQUIRE(dvis_conv_x3_image_bytes(N, C, H, W) < ((int64_t)1 << 31) && dvis_conv_x3_image_bytes(N, K, OH, OW) < ((int64_t)1 << 31),
               "dvis_conv_x3_image: images must stay below 2 GiB");
  return cx_launch(x, packed, bias, res, y, N, C, K, H, W, stride, taps, xexp, wexp, relu, stream, nullptr, 0, 0, 0, 1, image, oexp, ximg);
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
                     void *image, int oexp, const void *ximg) {
  DVIS_REQUIRE((x || ximg) && packed && (y || image), "dvis_conv1x1_x3: null operand");
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
  a.img = image, a.XG = (OW + 31) / 32, a.oscale = ldexpf(1.f, oexp), a.OH = OH;
  a.ximg = ximg, a.XG_in = (W + 31) / 32;
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
#define DVIS_CX_LAUNCH_IMG(NBV, IKV, INV, OUTV)                                                                                    \
  {                                                                                                                                \
    static DvisLdsOptIn opted;                                                                                                     \
    typedef Ring<2 * IKV * NBV / 8, 8, 8, (IKV == 4 ? 2 : 3)> R;                                                                   \
    a.tiles = (a.pixels + 255) / 256;                                                                                              \
    const size_t lds = (IKV == 4 ? 2 : 3) * R::kItemBytes;                                                                         \
    const int rc = dvis_lds_opt_in((const void *)conv1x1_x3_kernel<NBV, 8, IKV, INV, OUTV>, lds, &opted, "dvis_conv_x3_image");    \
    if (rc != DVIS_OK) return rc;                                                                                                  \
    hipLaunchKernelGGL((conv1x1_x3_kernel<NBV, 8, IKV, INV, OUTV>), dim3(grid), dim3(512), lds, st, a);                            \
    return dvis_check_launch("dvis_conv_x3_image");                                                                                \
  }
  if (ximg != nullptr || (image != nullptr && K != 64)) {      // operand images on either side (K = 64 out: the runtime branch of the plain kernels)
    DVIS_REQUIRE(x2 == nullptr, "dvis_conv_x3_image: no second source");
    if (K == 128) {
      if (ximg && image) DVIS_CX_LAUNCH_IMG(4, 2, true, true)
      if (ximg) DVIS_CX_LAUNCH_IMG(4, 2, true, false)
      DVIS_CX_LAUNCH_IMG(4, 2, false, true)
    }
    if (ximg && image) DVIS_CX_LAUNCH_IMG(8, 4, true, true)
    if (ximg) DVIS_CX_LAUNCH_IMG(8, 4, true, false)
    DVIS_CX_LAUNCH_IMG(8, 4, false, true)
  }
#undef DVIS_CX_LAUNCH_IMG
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
