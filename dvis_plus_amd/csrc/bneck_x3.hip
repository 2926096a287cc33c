// bneck_x3.hip — the res2 bottlenecks of the R50 (detectron2 BottleneckBlock with STRIDE_IN_1X1 False: 64 middle / 256 output
// channels on the stride-4 map) as a CHAIN: one kernel runs
//     conv2 (3x3, 64 -> 64) + ReLU  ->  conv3 (1x1, 64 -> 256) + shortcut + ReLU  ->  the NEXT block's conv1 (1x1, 256 -> 64 / 128) + ReLU
// on a wave's 32 pixels without the 64-channel map of conv2, or a second read of the 256-channel block output, ever touching
// memory.  At 720p the three layers of a block move 7.2 GB per 30-frame clip as separate launches (all three HBM-bound, 2.0 ms);
// the chain moves 4.5 GB: conv1's 64-channel map in (with its halo from L2), the shortcut in, the block output and the next
// conv1's map out.  Arithmetic and weight streaming are those of csrc/gemm_x3.hip / csrc/conv1x1_x3.hip (x3_common.h): every fp32
// operand as two f16 terms, three matrix-core products per pair, fp32 accumulation, one summation order per output (a frame's
// bits do not depend on its batch mates).
//
// What makes the chain possible:
//   * the k-order of an MFMA is free as long as both operands agree: conv3's weights are packed in ACCUMULATOR order
//     (x3_k(order = 1)), so the lane that holds conv2's output for its pixel holds conv3's "B" operand — and again for the next
//     conv1 behind conv3's epilogue (the FFN kernel's linear1 -> linear2 trick, two levels deep);
//   * the 64-channel map between two blocks is written as an OPERAND IMAGE, not as NCHW fp32: per 32-pixel group (n, y, x / 32)
//     8 KB = [k-step S][hi, lo][lane = 32 g + x % 32][8 halves] — exactly the fragment a wave's MFMA wants for the centre tap,
//     already split into its two f16 terms (same bytes as fp32).  The nine taps of conv2 are nine shifted reads of that image
//     (16 bytes per lane and k-step term; a row shift is another group, a column shift the neighbouring lane's slot; out of the
//     map = an out-of-range buffer offset = 0): no split arithmetic per tap, 8 loads per tap instead of 32;
//   * the identity shortcut is loaded straight into conv3's accumulator registers while conv2 runs (they are idle then) and
//     scaled into the accumulation's fixed point ((res + bias) * 2^(xexp + wexp) is exact): the epilogue has no loads left;
//   * a projection shortcut (the first block of the stage) is four more k-steps of conv3 over the block's input.
// Weight stream per tile (8 waves x 32 pixels): conv2 nine taps (144 KB) | conv3 (64 KB) | [shortcut (64 KB)] | next conv1
// (64 / 128 KB), through two 64 KB LDS slots with global_load_lds; one s_barrier per item.
#include "dvis_common.h"
#include "x3_common.h"

#pragma clang diagnostic ignored "-Winline-asm"      // (m0 on a clobber list: the LDS-DMA statement sets it)

namespace {

constexpr unsigned kBcOOB = 0x80000000u;   // beyond any served tensor (< 2 GiB): buffer loads return 0, stores are dropped
constexpr int kBcWaves = 8;
constexpr int kBcSlot = 65536;
constexpr int kBcTables = 2048;           // bytes of the shift tables in LDS (384 floats, padded)
constexpr int kBcGroup = 8192;             // bytes of one 32-pixel group of an operand image (64 channels)

struct BcArgs {
  const void *a1;              // operand image of conv1's output: groups x 8 KB
  const float *res;            // identity shortcut (N, 256, H, W), or NULL with the projection form
  const float *x2;             // projection form: the block's input (N, 64, H, W)
  const void *wp;              // packed stream (dvis_bneck_x3_pack)
  const float *b2, *b3, *b1;   // folded-BN shifts: conv2 (64), conv3 (+ shortcut) (256), next conv1 (64)
  float *y;                    // block output (N, 256, H, W)
  void *out;                   // next conv1's output as an operand image (TAIL)
  int N, H, W, XG;             // XG = ceil(W / 32) groups per row
  long long groups, tiles;     // N * H * XG; ceil(groups / 8)
  float xscale, inv2, inv3, s3, inv1;      // 2^xexp; 2^-(xexp + wexp) of conv2 / conv3 / next conv1; s3 = 1 / inv3
  int *flag;                   // range guard (x3_common.h)
  int tag;
};

// The stream of one tile.  conv3 runs in two halves of 128 output channels (the shortcut of a half lands in its accumulator
// registers while the previous stage multiplies), each followed by its share of the next conv1 (8 of its 16 k-steps):
//   identity, TAIL:   conv2 x 3 (48 KB each) | [conv3 h0 32 KB, tail h0 32 KB] | [conv3 h1, tail h1]
//   identity, !TAIL:  conv2 x 3 | [conv3 h0, conv3 h1]
//   projection, TAIL: conv2 x 3 | [conv3 h0 ++ shortcut h0: 8 k-steps, 64 KB] | tail h0 (32 KB) | [conv3 h1 ++ shortcut h1] | tail h1
template <bool DUAL, bool TAIL>
struct BcItems {
  static constexpr int kCount = DUAL ? 7 : TAIL ? 5 : 4;
  static constexpr int kConv2 = 9 * 16384;
  __host__ __device__ static constexpr int bytes(int i) {
    return i < 3 ? 49152 : DUAL ? ((i & 1) ? 65536 : 32768) : 65536;
  }
  __host__ __device__ static constexpr int off(int i) {
    int o = 0;
    for (int k = 0; k < i; ++k) o += bytes(k);
    return o;
  }
  __host__ __device__ static constexpr int pieces(int i) { return bytes(i) / (kBcWaves * kPiece); }
  static constexpr int kBytes = off(kCount);
};

template <bool DUAL, bool TAIL>
__global__ __launch_bounds__(kBcWaves * 64) void bneck_chain_kernel(const BcArgs a) {
  typedef BcItems<DUAL, TAIL> It;
  static_assert(!DUAL || TAIL, "the projection form is the first block of the stage: it always feeds a next block");
  extern __shared__ __attribute__((aligned(1024))) char lds_all[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  // the shift tables in FRONT of the two slots: their addresses fit a ds_read's 16-bit offset field (behind 128 KB every table
  // read wants its own address register)
  char *lds = lds_all + kBcTables;
  float *b3s = (float *)lds_all;                   // 256: (b3 [+ bs]) * s3
  float *b2s = b3s + 256;                          // 64
  float *b1s = b2s + 64;                           // 64
  for (int i = threadIdx.x; i < 256; i += kBcWaves * 64) b3s[i] = a.b3 ? a.b3[i] * a.s3 : 0.f;
  for (int i = threadIdx.x; i < 64; i += kBcWaves * 64) b2s[i] = a.b2 ? a.b2[i] : 0.f, b1s[i] = (TAIL && a.b1) ? a.b1[i] : 0.f;
  __syncthreads();

  // tiles: XCD x = blockIdx % 8 owns a contiguous eighth of them (neighbouring rows share their halo in that XCD's L2)
  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3, per = gridDim.x >> 3;
  const long long chunk = (a.tiles + 7) / 8;
  long long nloc = a.tiles - xcd * chunk;
  nloc = nloc < chunk ? nloc : chunk;
  const long long my = nloc > wi ? (nloc - wi + per - 1) / per : 0;
  if (my == 0) return;

  const long long HW = (long long)a.H * a.W;
  const unsigned cs = (unsigned)(HW * 4);                    // bytes between two channels of a pixel (fp32 NCHW maps)
  const __amdgpu_buffer_rsrc_t ra = dvis_make_rsrc_uniform(a.a1, (unsigned)(a.groups * kBcGroup));
  const __amdgpu_buffer_rsrc_t rr = dvis_make_rsrc_uniform(a.res ? (const void *)a.res : (const void *)a.y, (unsigned)(a.N * 256 * HW * 4));
  const __amdgpu_buffer_rsrc_t ry = dvis_make_rsrc_uniform(a.y, (unsigned)(a.N * 256 * HW * 4));
  const __amdgpu_buffer_rsrc_t rx2 = dvis_make_rsrc_uniform(DUAL ? (const void *)a.x2 : (const void *)a.y, DUAL ? (unsigned)(a.N * 64 * HW * 4) : 0u);
  const __amdgpu_buffer_rsrc_t ro = dvis_make_rsrc_uniform(TAIL ? a.out : (void *)a.y, TAIL ? (unsigned)(a.groups * kBcGroup) : 0u);

  // ---- LDS-DMA of the weight stream: an item goes to the slot the previous one does not occupy; every wave moves pieces
  // wave, wave + 8, ... (scalar base + the lane's 16 bytes: no per-piece address registers)
  // MUBUF form written as inline assembly (`buffer_load_dwordx4 ... lds`): hipcc's wait-count insertion treats the FLAT-encoded
  // global_load_lds as "may touch LDS and memory" and from then on turns every vmcnt / lgkmcnt wait into a full drain — the
  // taps' operand loads would wait for the shortcut rows and the pieces requested behind them.  The assembler statement is not
  // counted at all: the compiler's counted waits for ITS loads stay valid (vmcnt only over-waits by the pieces in flight), and
  // the pieces themselves are waited for explicitly (begin_item).
  const dvis_v4u rw = {(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)a.wp),
                       (unsigned)__builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)a.wp >> 32)), (unsigned)It::kBytes, 0x00020000u};
  const unsigned lane16 = lane * 16, wlane = lane16 + wave * kPiece;
  const unsigned lds0 = (unsigned)(uintptr_t)(DVIS_LDS char *)lds + wave * kPiece;
  int par = 0;                                               // slot of the item being multiplied
  auto piece = [&](int item, int slot, int i) {
    if (i < It::pieces(item)) {
      const unsigned dst = lds0 + slot * kBcSlot + i * (kBcWaves * kPiece), so = It::off(item) + i * (kBcWaves * kPiece);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(wlane), "s"(rw), "s"(so) : "memory", "m0");
    }
  };
  // the item is complete in LDS for every wave; returns its slot and points `par` at the slot of the next one
  auto begin_item = [&]() -> const char * {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces (and everything older)
    __builtin_amdgcn_s_barrier();
    const char *stage = lds + par * kBcSlot;
    par ^= 1;
    return stage;
  };

  // ---- geometry of the wave's group in tile t
  struct Geo {
    unsigned row[3], col[3];     // operand-image offsets of the rows y - 1 .. y + 1 (group 0 of the row) and of the columns x - 1 .. x + 1 within a row; kBcOOB outside
    unsigned pix;                // byte offset of (n, channel 4 g, y, x) in a 256-channel fp32 map; kBcOOB for a pixel outside
    unsigned pix64;              // of (n, channel 8 g, y, x) in a 64-channel map (projection form's input)
    unsigned img;                // byte offset of the lane's 16 bytes in the group of an operand image; kBcOOB for a group past the end
  };
  auto geometry = [&](long long t) {
    Geo q;
    const long long G = t * kBcWaves + wave;
    const bool gok = G < a.groups;
    const long long rowi = G / a.XG;                         // n * H + y
    const int xg = (int)(G - rowi * a.XG);
    const long long n = rowi / a.H;
    const int y = (int)(rowi - n * a.H), x = 32 * xg + j;
    const bool pok = gok && x < a.W;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int yy = y + d - 1, xx = x + d - 1;
      q.row[d] = gok && yy >= 0 && yy < a.H ? (unsigned)((rowi + d - 1) * a.XG * kBcGroup) : kBcOOB;
      q.col[d] = xx >= 0 && xx < a.W ? (unsigned)((xx >> 5) * kBcGroup + g * 512 + (xx & 31) * 16) : kBcOOB;
    }
    // (the lane's half g is part of the per-lane offset — channel 4 g of the accumulator order, 8 g of the natural one — so
    // that the channel term of every access is wave-uniform and travels in a scalar register)
    const unsigned sp = (unsigned)((y * (long long)a.W + x) * 4);
    q.pix = pok ? (unsigned)(n * 256 * HW * 4) + sp + 4 * g * cs : kBcOOB;
    q.pix64 = pok ? (unsigned)(n * 64 * HW * 4) + sp + 8 * g * cs : kBcOOB;
    q.img = gok ? (unsigned)(G * kBcGroup) + lane16 : kBcOOB;
    return q;
  };
  // the "B" operand of tap `tap` (4 k-steps x (hi, lo)): 8 loads of 16 bytes
  auto load_tap = [&](const Geo &q, int tap, h8 *bh, h8 *bl) {
    const unsigned r = q.row[tap / 3], c = q.col[tap % 3];
    const unsigned off = (r | c) & kBcOOB ? kBcOOB : r + c;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bh[s] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 2048 * s, 0));
      bl[s] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 2048 * s + 1024, 0));
    }
  };
  // 16 values of the lane's accumulator block (registers 8 u + e of k-step u) -> the operand pair of the two k-steps
  auto split_block = [&](const float *v, float s, h8 *hi, h8 *lo) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f4 p0 = {v[8 * u], v[8 * u + 1], v[8 * u + 2], v[8 * u + 3]}, p1 = {v[8 * u + 4], v[8 * u + 5], v[8 * u + 6], v[8 * u + 7]};
      split8(p0, p1, s, hi[u], lo[u]);
    }
  };

  long long tile = xcd * chunk + wi;
  Geo gm = geometry(tile);
  // prologue: item 0 of the first tile, its first tap
#pragma unroll
  for (int i = 0; i < It::pieces(0); ++i) piece(0, 0, i);
  h8 B0h[4], B0l[4];                                         // tap 0's operand travels from the end of one tile to the next
  load_tap(gm, 0, B0h, B0l);

  for (long long w = 0; w < my; ++w) {
    const bool more = w + 1 < my;

    // ================= phase A: conv2, nine taps of 24 products; the first half of the identity shortcut lands in acc3 meanwhile
    f16v acc2[2], acc3[8];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc2[nb][i] = 0.f;
    float raw[32];                                           // projection form: 64 input channels of the lane's pixel
    h8 Bh[2][4], Bl[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) Bh[0][s] = B0h[s], Bl[0][s] = B0l[s];
    // shortcut rows of conv3's blocks [nb0, nb0 + n): requests into the accumulator registers
    // (`csl`: the channel stride made opaque at each use — as a loop invariant, its 128 multiples are hoisted out of the tile
    // loop into scalar registers the kernel does not have)
    auto load_res = [&](int nb0, int n) {
      unsigned csl = cs;
      asm volatile("" : "+s"(csl));
#pragma unroll
      for (int r = 16 * nb0; r < 16 * (nb0 + n); ++r) {
        const int nb = r >> 4, q = (r >> 2) & 3, i = r & 3;
        acc3[nb][4 * q + i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, gm.pix, (unsigned)(32 * nb + 8 * q + i) * csl, 0));
      }
    };
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const char *stage = begin_item();
      const int slot = par;
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        const int t = 3 * it + tt, cur = t & 1;
        // requests first: the shortcut rows (taps 0 and 1: their latency is covered by the rest of conv2), the next tap's operand
        if (!DUAL && t < 2) load_res(2 * t, 2);
        if (DUAL && t == 6) {
          unsigned csl = cs;
          asm volatile("" : "+s"(csl));
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              raw[8 * s + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx2, gm.pix64, (unsigned)(16 * s + e) * csl, 0));
        }
        if (t + 1 < 9) load_tap(gm, t + 1, Bh[cur ^ 1], Bl[cur ^ 1]);
        // the next item's pieces in the first two taps of this one (4 slots each; a 48 KB item leaves the last two empty)
        if (tt < 2)
          mma_item<4, 2, 4>(stage + tt * 16384, lane, acc2, Bh[cur], Bl[cur], [&](int i) { piece(it + 1, slot, 4 * tt + i); });
        else
          mma_item<4, 2>(stage + tt * 16384, lane, acc2, Bh[cur], Bl[cur]);
      }
    }

    // ================= conv2's output (in accumulator order) = conv3's operand [+ the block input for the projection shortcut]
    h8 xh[DUAL ? 8 : 4], xl[DUAL ? 8 : 4];
    {
      float v[2][16];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f4 b = *(const f4 *)(b2s + 32 * nb + 8 * q + 4 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[nb][4 * q + i] = fmaxf(acc2[nb][4 * q + i] * a.inv2 + b[i], 0.f);
        }
      split_block(v[0], a.xscale, xh, xl);
      split_block(v[1], a.xscale, xh + 2, xl + 2);
    }

    // ================= phases B / C per half of conv3's output channels
    f16v acc1[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc1[nb][i] = 0.f;
    float chk = 0.f;                   // range guard: NaN as soon as one value is non-finite BEFORE its ReLU
    // items behind conv2: identity + tail: 3 + h; identity without a tail: 3 (both halves); projection: conv3 3 + 2 h, tail 4 + 2 h
    auto next_of = [&](int i) { return i + 1 == It::kCount ? 0 : i + 1; };
    const char *stage = nullptr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int item = DUAL ? 3 + 2 * h : TAIL ? 3 + h : 3;
      if (h == 0 || DUAL || TAIL) stage = begin_item();      // (identity without a tail: both halves in one item)
      const int slot = par;
      const char *c3w = stage + ((!DUAL && !TAIL && h == 1) ? 32768 : 0);
      if (!DUAL && h == 0) load_res(4, 4);                   // the second half's shortcut rows: they land under this half
      if (DUAL && h == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const f4 lo4 = {raw[8 * s], raw[8 * s + 1], raw[8 * s + 2], raw[8 * s + 3]};
          const f4 hi4 = {raw[8 * s + 4], raw[8 * s + 5], raw[8 * s + 6], raw[8 * s + 7]};
          split8(lo4, hi4, a.xscale, xh[4 + s], xl[4 + s]);
        }
      }
      // acc3 = (shortcut + bias) * s3: the half's first block in front, block t + 1 behind the first product of block t
      auto init = [&](int nb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f4 b = *(const f4 *)(b3s + 32 * nb + 8 * q + 4 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc3[nb][4 * q + i] = DUAL ? b[i] : __builtin_fmaf(acc3[nb][4 * q + i], a.s3, b[i]);
        }
      };
      init(4 * h);
      // the next item's pieces (none under the first half of a one-item conv3)
      const int nxt = next_of(item);
      const bool nxt_on = (DUAL || TAIL || h == 1) && (nxt > 0 || more);
      mma_item<DUAL ? 8 : 4, 4, 8>(c3w, lane, acc3 + 4 * h, xh, xl, [&](int i) {
        if (nxt_on) piece(nxt, slot, i);
      }, [&](int t) {
        if (t < 3) init(4 * h + t + 1);
      });
      const char *tw = c3w + 32768;                          // identity form: the tail's half follows conv3's in the item
      int tslot = slot;
      if (DUAL) {
        tw = begin_item();
        tslot = par;
      }
      // ---- conv3's epilogue block by block; each block's 32 channels = two k-steps of the next conv1
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int nb = 4 * h + b;
        float v[16];
        unsigned csl = cs;
        asm volatile("" : "+s"(csl));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float t = acc3[nb][r] * a.inv3;
          chk = __builtin_fmaf(t, 0.f, chk);
          v[r] = fmaxf(t, 0.f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[r]), ry, gm.pix, (unsigned)(32 * nb + 8 * (r >> 2) + (r & 3)) * csl, 0);
        }
        asm volatile("" : "+v"(chk));      // (taken here: hipcc otherwise sinks the guard's 128 multiply-adds, and their operands, to the end of the tile)
        if constexpr (TAIL) {
          h8 th[2], tl[2];
          split_block(v, a.xscale, th, tl);
          if (DUAL) {
            const int n2 = next_of(item + 1);
            const bool on2 = n2 > 0 || more;
            mma_item<2, 2, 2>(tw + b * 8192, lane, acc1, th, tl, [&](int i) {
              if (on2) piece(n2, tslot, 2 * b + i);
            });
          } else {
            mma_item<2, 2>(tw + b * 8192, lane, acc1, th, tl);
          }
        }
      }
    }

    // ================= phase D: the next conv1's epilogue -> operand image
    if constexpr (TAIL) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f4 b = *(const f4 *)(b1s + 32 * nb + 8 * q + 4 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float t = acc1[nb][4 * q + i] * a.inv1 + b[i];
            chk = __builtin_fmaf(t, 0.f, chk);
            v[4 * q + i] = fmaxf(t, 0.f);
          }
        }
        h8 oh[2], ol[2];
        split_block(v, a.xscale, oh, ol);
        store_fragments4(ro, gm.img, oh[0], 4096 * nb, ol[0], 4096 * nb + 1024, oh[1], 4096 * nb + 2048, ol[1], 4096 * nb + 3072);
      }
    }
    if (a.flag != nullptr && chk != chk && gm.pix != kBcOOB) atomicCAS(a.flag, 0, a.tag);
    // the next tile's first tap
    tile += per;
    gm = geometry(more ? tile : a.tiles);                    // (a.tiles: every offset out of range)
    load_tap(gm, 0, B0h, B0l);                               // (always: past the last tile every offset is out of range, the loads return 0)
  }
}

// (64, 64, 3, 3) / (K, C) weights -> the stream (BcItems).  One thread per (2 KB fragment pair, lane), x3_pack_kernel's fragment layout.
struct BpArgs {
  const float *w2, *w3, *ws, *w1;
  int tail, dual;
  float s2, s3, s1;
  _Float16 *out;
};
__global__ void bneck_pack_kernel(const BpArgs p, int fragments) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= fragments * 64) return;
  const int lane = idx & 63, n = lane & 31, g = lane >> 5;
  int f = idx >> 6;
  _Float16 *o = p.out + (size_t)f * 1024 + lane * 8;
  if (f < 72) {                      // conv2: [k-step of (tap, channel)][block]; k = 64 tap + channel, channels in accumulator order
    x3_pack_fragment(p.w2, 576, 64, 576, 32 * (f % 2) + n, 1, f / 2, g, p.s2, o, o + 512, 9);
    return;
  }
  f -= 72;
  const int c3f = p.dual ? 32 : 16, tf = p.tail ? 16 : 0;      // fragments of a half: conv3 (+ shortcut), tail
  const int h = f / (c3f + tf);
  f -= h * (c3f + tf);
  if (f < c3f) {                     // conv3 half h: [k-step][block 4 h .. 4 h + 3]; k-steps 4 .. 7 = the projection shortcut (natural order)
    const int S = f / 4, nb = 4 * h + f % 4;
    if (S < 4)
      x3_pack_fragment(p.w3, 64, 256, 64, 32 * nb + n, 1, S, g, p.s3, o, o + 512);
    else
      x3_pack_fragment(p.ws, 64, 256, 64, 32 * nb + n, 0, S - 4, g, p.s3, o, o + 512);
    return;
  }
  f -= c3f;                          // next conv1, k-steps 8 h .. 8 h + 7 (K = 256 in accumulator order): [k-step][block]
  x3_pack_fragment(p.w1, 256, 64, 256, 32 * (f % 2) + n, 1, 8 * h + f / 2, g, p.s1, o, o + 512);
}

}  // namespace

DVIS_EXPORT int64_t dvis_bneck_x3_packed_bytes(int tail, int dual) {
  if (dual && !tail) return -1;
  return dual ? BcItems<true, true>::kBytes : tail ? BcItems<false, true>::kBytes : BcItems<false, false>::kBytes;
}

DVIS_EXPORT int64_t dvis_bneck_x3_image_bytes(int64_t N, int H, int W) {
  if (N < 0 || H <= 0 || W <= 0) return -1;
  return N * H * ((W + 31) / 32) * (int64_t)kBcGroup;
}

/* w2 (64, 64, 3, 3), w3 (256, 64), ws (256, 64) or NULL, w1 (64, 256) or NULL: folded weights; e2 / e3 / e1: their scaling
 * exponents (ws shares e3: the two accumulate together) */
DVIS_EXPORT int dvis_bneck_x3_pack(const float *w2, const float *w3, const float *ws, const float *w1, int e2, int e3, int e1,
                                   void *packed, void *stream) {
  DVIS_REQUIRE(w2 && w3 && packed, "dvis_bneck_x3_pack: null operand");
  DVIS_REQUIRE(!ws || w1, "dvis_bneck_x3_pack: the projection form needs the next block's conv1");
  DVIS_REQUIRE(e2 >= -60 && e2 <= 60 && e3 >= -60 && e3 <= 60 && e1 >= -60 && e1 <= 60, "dvis_bneck_x3_pack: exponents %d %d %d", e2, e3, e1);
  BpArgs p = {w2, w3, ws, w1, w1 != nullptr, ws != nullptr, ldexpf(1.f, e2), ldexpf(1.f, e3), ldexpf(1.f, e1), (_Float16 *)packed};
  const int fragments = (int)(dvis_bneck_x3_packed_bytes(p.tail, p.dual) / 2048);
  hipLaunchKernelGGL(bneck_pack_kernel, dim3((fragments * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, fragments);
  return dvis_check_launch("dvis_bneck_x3_pack");
}

DVIS_EXPORT int dvis_bneck_x3_supported(int64_t N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  if (N * 256 * H * W * 4 >= ((int64_t)1 << 31)) return 0;          // 32-bit buffer offsets
  if (dvis_bneck_x3_image_bytes(N, H, W) >= ((int64_t)1 << 31)) return 0;
  return 1;
}

/* One res2 bottleneck from conv1's output on, and (out != NULL) the next block's conv1:
 *   a2 = relu(conv3x3(a1) + b2);  y = relu(conv1x1(a2, w3) + b3 + (res | conv1x1(x2, ws)));  out = relu(conv1x1(y, w1) + b1)
 * a1, out: operand images (dvis_bneck_x3_image_bytes; written by dvis_conv1x1_x3_image or by a previous call);
 * res: (N, 256, H, W) identity shortcut, or NULL with x2 (N, 64, H, W) and a stream packed with ws. */
DVIS_EXPORT int dvis_bneck_x3(const void *a1, const float *res, const float *x2, const void *packed, const float *b2, const float *b3,
                              const float *b1, float *y, void *out, int N, int H, int W, int xexp, int e2, int e3, int e1, void *stream) {
  DVIS_REQUIRE(a1 && packed && y, "dvis_bneck_x3: null operand");
  DVIS_REQUIRE((res != nullptr) != (x2 != nullptr), "dvis_bneck_x3: exactly one of res (identity shortcut) and x2 (projection shortcut's input)");
  DVIS_REQUIRE(!x2 || out, "dvis_bneck_x3: the projection form always feeds a next block");
  DVIS_REQUIRE(dvis_bneck_x3_supported(N, H, W), "dvis_bneck_x3: (N %d, %d x %d) is not served (tensors below 2 GiB)", N, H, W);
  DVIS_REQUIRE((uintptr_t)a1 % 16 == 0 && (uintptr_t)packed % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)y % 4 == 0,
               "dvis_bneck_x3: images and the stream must be 16-byte aligned");
  BcArgs a = {};
  a.a1 = a1, a.res = res, a.x2 = x2, a.wp = packed, a.b2 = b2, a.b3 = b3, a.b1 = b1, a.y = y, a.out = out;
  a.N = N, a.H = H, a.W = W, a.XG = (W + 31) / 32;
  a.groups = (long long)N * H * a.XG, a.tiles = (a.groups + kBcWaves - 1) / kBcWaves;
  a.xscale = ldexpf(1.f, xexp), a.inv2 = ldexpf(1.f, -(xexp + e2)), a.inv3 = ldexpf(1.f, -(xexp + e3)), a.s3 = ldexpf(1.f, xexp + e3);
  a.inv1 = ldexpf(1.f, -(xexp + e1));
  const X3Guard gd = dvis_x3_guard();
  a.flag = gd.flag, a.tag = gd.tag;
  const int grid = dvis_x3_persistent_cus();
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = 2 * kBcSlot + kBcTables;
#define DVIS_BC_LAUNCH(DUALV, TAILV)                                                                              \
  {                                                                                                               \
    static DvisLdsOptIn opted;                                                                                    \
    const int rc = dvis_lds_opt_in((const void *)bneck_chain_kernel<DUALV, TAILV>, lds, &opted, "dvis_bneck_x3"); \
    if (rc != DVIS_OK) return rc;                                                                                 \
    hipLaunchKernelGGL((bneck_chain_kernel<DUALV, TAILV>), dim3(grid), dim3(kBcWaves * 64), lds, st, a);          \
  }
  if (x2) DVIS_BC_LAUNCH(true, true) else if (out) DVIS_BC_LAUNCH(false, true) else DVIS_BC_LAUNCH(false, false)
#undef DVIS_BC_LAUNCH
  return dvis_check_launch("dvis_bneck_x3");
}
