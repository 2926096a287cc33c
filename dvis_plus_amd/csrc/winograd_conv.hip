// 3x3 / stride 1 / pad 1 convolution, NCHW fp32, as Winograd F(2x2, 3x3) on the fp32 matrix cores (round 3).
//
// What it replaces: the library's kernels for the 3x3 convolutions of the path — the pixel decoder's FPN output
// convolution (mask2former/modeling/pixel_decoder/msdeformattn.py:262-270 `output_conv`, 256 -> 256 at stride 4: 2.08 TFLOP per
// 30-frame clip, 16.8 ms in MIOpen's implicit-GEMM kernel = the fp32-MFMA peak for 9 multiplies per output) and conv2 of
// the R50 bottlenecks (detectron2 BottleneckBlock; MIOpen runs those as a VALU Winograd).  F(2x2, 3x3)
// needs 4 multiplies per output instead of 9 and the 16 transform-domain products are plain GEMMs
//     M_xi[k][tile] = sum_c U_xi[k][c] V_xi[c][tile],   xi = 0..15,
// which v_mfma_f32_16x16x4_f32 does exactly in fp32: 2.25x fewer matrix-core cycles than any direct formulation.
//
// Workgroup = 64 consecutive 2x2-output tiles (in (n, ty, tx) order — no spatial blocking, so no padding waste at
// 23 x 40) x 64 output channels, 8 waves.  Wave w owns the 16 output channels kb16 = w & 3, ALL 64 tiles and HALF of the
// transform positions (xi in [8 half, 8 half + 8), half = w >> 2): 8 x 4 accumulator tiles = 128 VGPRs, two waves per SIMD.
//   * U (the transformed weights, packed once by dvis_conv3x3_winograd_pack) goes from L2 straight into the MFMA A layout:
//     each wave reads only ITS channels and positions, 4 KB per 8-channel stage in four fully coalesced 16-byte loads — no
//     LDS, no redundancy between the waves.
//   * V (the transformed input) is made in the workgroup: wave w loads the 4x4 input patches of channel w of the stage for
//     the 64 tiles (lane = tile; out-of-image elements carry an out-of-range buffer offset and read 0: exact zero padding,
//     no branches), applies B^T d B (32 adds) and writes the 16 positions to LDS, 64 consecutive dwords per row; the B
//     operands of 4 MFMAs are one ds_read_b128 (accumulator tile tb of lane column j = tile 4 j + tb).
//   * LDS is double-buffered (2 x 32 KB): one barrier per stage of 8 input channels = per 64 MFMAs of a wave.  The transform
//     of the NEXT stage's patch is threaded through the stage's MFMAs in 32 micro-slices (see `stage`); patches are requested
//     two stages ahead, U one stage ahead.
//   * What bounds it (timing ablations, profiles/r03_winograd_ablation*.txt): MFMAs alone 6.4 ms at the FPN shape = the matrix
//     pipe at 100 %; everything else alone 1.8 ms; together 8.4 ms — the SUM, whatever the schedule (separate phases with the
//     two waves of a SIMD in opposite order, 6 slices, 32 micro-slices, patches one or two stages ahead: 8.39 - 8.47 ms).  The
//     fp32 MFMA and the fp32 VALU do not overlap on a SIMD here, so every VALU instruction of the transform is paid in full —
//     which is why the edge tiles are no longer shifted in registers (40 of ~80 VALU per wave and stage) but loaded as they lie
//     and masked (8): 8.4 -> 7.95 ms.
//   * Output transform A^T M A is linear, so each half reduces ITS 8 positions to a partial 2x2 block in registers and the
//     halves exchange only those (64 KB through the LDS that the stages no longer need) — not the 16 M tiles.
// Accumulation order over input channels is fixed (stages in order, no split-K, no atomics): bit-reproducible.
#include "dvis_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kTiles = 64;                 // tiles per workgroup
constexpr int kKw = 64;                    // output channels per workgroup
constexpr int kCc = 8;                     // input channels per stage (two MFMA k-steps)
constexpr int kRow = 64;                   // floats per channel row of V in LDS: tile t at dword t
constexpr int kPos = kCc * kRow;           // floats per transform position in a stage
constexpr int kStage = 16 * kPos;          // floats per stage (32 KB)
constexpr unsigned kOOB = 0x80000000u;

struct WinoArgs {
  const float *x, *uf, *bias;
  float *y;
  int N, C, K, H, W, TY, TX, relu, nsp;
  long long tiles;
};

template <int ABL>   // ABL != 0 (-DDVIS_WINO_ABLATION builds, tools/exp/wino_abl.sh): 1 no patch loads, 2 no transform, 4 no U loads, 8 no MFMAs
__global__ __launch_bounds__(512, 2) void winograd_f2x3_kernel(const WinoArgs a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (provably wave-uniform: scalar offsets, no waterfall loops)
  const int j = lane & 15, g = lane >> 4;
  const int kb16 = wv & 3, half = wv >> 2;
  // blocks that share input tiles (the K / 64 channel blocks of one tile group) run on the SAME XCD, back to back: block
  // b lands on XCD b % 8, so the tile group takes the low 3 bits and the channel block the next ones.
  const int KB = a.K / kKw;
  const int grp = blockIdx.x / (8 * KB), rem = blockIdx.x - grp * 8 * KB;
  const int kb = rem >> 3, sp = grp * 8 + (rem & 7);
  if (sp >= a.nsp) return;
  const long long p0 = (long long)sp * kTiles;
  const int per_img = a.TY * a.TX;
  const int n0 = (int)(p0 / per_img);   // per_img >= 64: the workgroup's tiles lie in images n0 and n0 + 1
  const long long plane = (long long)a.H * a.W, img = plane * a.C;
  const int nch = a.C / kCc;

  // ---- transform role: lane = tile, wave = channel of the stage.  Patch element (i, jj) = input (2 ty - 1 + i, 2 tx - 1 + jj).
  // A patch row is ONE 16-byte load (4-byte aligned): 16 dword loads per lane and stage kept the texture-address unit busy
  // for 2500 of a stage's 6100 cycles and were not hidden (timing ablation: 10.4 ms -> 7.0 without them; profiles/
  // r03_winograd_ablation.txt).  A row outside the image (or a tile past the end) carries an out-of-range offset and reads
  // 0.  The leftmost tile of a row starts at column -1 and the rightmost (W even) ends at column W.  The main loop loads
  // exactly those 16 bytes — the stray element is the neighbouring row's pixel, or lies past the tensor's end where the
  // descriptor's per-dword range check returns 0 — and clears it with one AND per row end (2 VALU per row; shifting edge
  // tiles in registers cost 10).  Its offsets are biased by +4 against a base 4 bytes in front of the window, because a
  // NEGATIVE buffer offset zeroes the whole load (tools/exp/struct_probe/raw_probe.hip).  The one load that would start 4
  // bytes in front of the TENSOR (tile 0 of image 0 in channel 0) belongs to stage 0, which the prologue handles with loads
  // that stay inside the image row (edge tiles load columns 0..3 resp. W-4..W-1 and shift in registers: `rowq0`, `rx0`).
  unsigned rowq[4], rowq0[4];
  bool left, right;
  {
    const long long p = p0 + lane;
    const bool pv = p < a.tiles;
    const int pi = pv ? (int)(p - (long long)n0 * per_img) : 0;
    const int nn = pi / per_img, r = pi - nn * per_img;
    const int ty = r / a.TX, tx = r - ty * a.TX;
    left = tx == 0, right = 2 * tx + 2 >= a.W;
    const int x0 = left ? 0 : (right ? a.W - 4 : 2 * tx - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int yy = 2 * ty - 1 + i;
      const bool ok = pv && yy >= 0 && yy < a.H;
      const long long row = (long long)nn * img + (long long)yy * a.W;
      rowq0[i] = ok ? (unsigned)((row + x0) * 4) : kOOB;
      rowq[i] = ok ? (unsigned)((row + 2 * tx - 1) * 4 + 4) : kOOB;
    }
  }
  const int n_here = min(2, a.N - n0);
  const __amdgpu_buffer_rsrc_t rx0 = dvis_make_rsrc_uniform(a.x + (long long)n0 * img, (unsigned)(n_here * img * 4));
  const __amdgpu_buffer_rsrc_t rx = dvis_make_rsrc_uniform(reinterpret_cast<const char *>(a.x + (long long)n0 * img) - 4,
                                                           (unsigned)(n_here * img * 4 + 4));
  const __amdgpu_buffer_rsrc_t ru = dvis_make_rsrc_uniform(a.uf, (unsigned)(16ll * a.K * a.C * 4));
  const unsigned plane_bytes = (unsigned)(plane * 4);
  const unsigned u_lane = (unsigned)lane * 16u;
  const unsigned u_blk = (unsigned)((kb * 4 + kb16) * nch);

  auto load_d = [&](int ch, dvis_f4 (&d)[4]) {
    if constexpr (ABL & 1) return;
    const unsigned so = (unsigned)(ch * kCc + wv) * plane_bytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, rowq[i], so, 0));
  };
  auto load_u = [&](int ch, dvis_f4 (&u)[4]) {
    if constexpr (ABL & 4) return;
    const unsigned so = ((u_blk + (unsigned)ch) * 2u + (unsigned)half) * 4096u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      u[q] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(ru, u_lane + 1024u * q, so, 0));
  };
  // edge tiles loaded their rows one column to the right (`left`) / left (`right`) of the patch: shift, zero the column
  // outside the image.  Bit masks, not `c ? a : b`: hipcc turned the nested selects into exec-masked branches.
  const unsigned lm = left ? ~0u : 0u, rm = right ? ~0u : 0u, nm = ~(lm | rm);
  auto shift_row = [&](const dvis_f4 &q, float &e0, float &e1, float &e2, float &e3) {
    // (by value through __float_as_uint: __builtin_bit_cast applied to the element lvalue q[k] reads element 0 for every k
    // with this hipcc — the whole row collapsed to its first float)
    const unsigned q0 = __float_as_uint(q[0]), q1 = __float_as_uint(q[1]), q2 = __float_as_uint(q[2]),
                   q3 = __float_as_uint(q[3]);
    e0 = __uint_as_float((q0 & nm) | (q1 & rm));
    e1 = __uint_as_float((q0 & lm) | (q1 & nm) | (q2 & rm));
    e2 = __uint_as_float((q1 & lm) | (q2 & nm) | (q3 & rm));
    e3 = __uint_as_float((q2 & lm) | (q3 & nm));
  };
  // V = B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], position xi = 4 i + jj, to row (xi, channel wv)
  auto transform_store = [&](const dvis_f4 (&q)[4], float *stage) {
    if constexpr (ABL & 2) return;
    float *vw = stage + wv * kRow + lane;   // 64 lanes, 64 consecutive dwords: conflict-free under the (a/4) % 32 write banking
    float d[16], t[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) shift_row(q[i], d[4 * i], d[4 * i + 1], d[4 * i + 2], d[4 * i + 3]);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      t[jj] = d[jj] - d[8 + jj];
      t[4 + jj] = d[4 + jj] + d[8 + jj];
      t[8 + jj] = d[8 + jj] - d[4 + jj];
      t[12 + jj] = d[4 + jj] - d[12 + jj];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vw[(4 * i) * kPos] = t[4 * i] - t[4 * i + 2];
      vw[(4 * i + 1) * kPos] = t[4 * i + 1] + t[4 * i + 2];
      vw[(4 * i + 2) * kPos] = t[4 * i + 2] - t[4 * i + 1];
      vw[(4 * i + 3) * kPos] = t[4 * i + 1] - t[4 * i + 3];
    }
  };

  dvis_f4 acc[8][4];
#pragma unroll
  for (int x8 = 0; x8 < 8; ++x8)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb) acc[x8][tb] = dvis_f4{0.f, 0.f, 0.f, 0.f};
  // u[q] = {U(xi0, s0), U(xi0, s1), U(xi0 + 1, s0), U(xi0 + 1, s1)}, xi0 = 8 half + 2 q; k-step s = channels 4 s + g
  // B operands: one ds_read_b128 per (position, k-step) = tiles 4 j .. 4 j + 3 of the row: accumulator tile tb holds the 16
  // tiles {4 j + tb}.  Rows are NOT padded: ds_read_b128 is serviced in four 16-lane groups that each take j = 0..15 once
  // (from two different lane groups g), so equal bank alignment of the rows is what makes it conflict-free — padded to 80
  // floats the groups collided 2-way (SQ_LDS_BANK_CONFLICT 12 % of the CU cycles), and a tile-block-major row ("tile
  // 16 tb + j") made every ds_write_b32 2-way ((a/4) % 32 banking over 32-lane halves).  The reads of position x8 + 1 are
  // issued before the 8 MFMAs of position x8 (hipcc otherwise reads each operand right in front of its MFMA and waits).
  float *s0 = lds, *s1 = lds + kStage;
  dvis_f4 d[4] = {}, ua[4] = {}, ub[4] = {};
  load_u(0, ua);
  {   // stage 0 (channel wv of the image): loads that stay inside the image row, edge tiles shifted in registers
    const unsigned so = (unsigned)wv * plane_bytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rx0, rowq0[i], so, 0));
  }
  transform_store(d, s0);
  // One stage = the 64 MFMAs of the wave on the current buffer WITH the transform of the next stage's patch threaded through
  // them in 32 micro-slices: after every PAIR of MFMAs at most 7 other instructions (one work item of <= 6 VALU / LDS-write /
  // VMEM instructions, sometimes a B-operand read).  Why that fine: the matrix pipe of a SIMD is fed in order by its two
  // waves, and a wave hides only ~7 single-issue instructions behind one of its MFMAs (32 cycles); a longer run of VALU is a
  // hole in ITS MFMA stream, and the partner — same code, released by the same barrier — has its hole at the same time.
  // Measured: separate phases (MFMAs, then transform; the two waves of a SIMD in opposite order) ran at the SUM of the parts
  // (MFMAs alone 6.4 ms, loads + transform alone 2.1 ms, together 8.4 ms), and so did 6 slices of ~28 VALU each.
  // Items, in dependency order:  S(i, h) half of patch row i, its out-of-image end cleared (1) x 8;  T(jj) column jj of B^T d (4) x 4;  V(i) row i of
  // (B^T d) B + its 4 positions to LDS (4 + 2 write2) x 4;  L(i) request row i of the patch two stages ahead x 4.
  auto fence = [] { __builtin_amdgcn_sched_barrier(0); };
  auto stage = [&](const float *cur, const dvis_f4 (&u)[4], float *nxt, dvis_f4 (&d)[4], int ch_load) {
    const dvis_f4 *vr = reinterpret_cast<const dvis_f4 *>(cur + (half * 8) * kPos + g * kRow + 4 * j);
    float *vw = nxt + wv * kRow + lane;
    const unsigned so_load = (unsigned)(ch_load * kCc + wv) * plane_bytes;
    dvis_f4 b[2][2];
    float e[16], t[16];
    auto S = [&](int i, int h) {
      const unsigned q0 = __float_as_uint(d[i][0]), q1 = __float_as_uint(d[i][1]), q2 = __float_as_uint(d[i][2]),
                     q3 = __float_as_uint(d[i][3]);
      if (h == 0) e[4 * i] = __uint_as_float(q0 & ~lm), e[4 * i + 1] = d[i][1];     // column -1 of the leftmost tile
      else e[4 * i + 2] = d[i][2], e[4 * i + 3] = __uint_as_float(q3 & ~rm);       // column W of the rightmost tile
      (void)q1, (void)q2;
    };
    auto T = [&](int jj) {
      t[jj] = e[jj] - e[8 + jj];
      t[4 + jj] = e[4 + jj] + e[8 + jj];
      t[8 + jj] = e[8 + jj] - e[4 + jj];
      t[12 + jj] = e[4 + jj] - e[12 + jj];
    };
    auto V = [&](int i) {
      vw[(4 * i) * kPos] = t[4 * i] - t[4 * i + 2];
      vw[(4 * i + 1) * kPos] = t[4 * i + 1] + t[4 * i + 2];
      vw[(4 * i + 2) * kPos] = t[4 * i + 2] - t[4 * i + 1];
      vw[(4 * i + 3) * kPos] = t[4 * i + 1] - t[4 * i + 3];
    };
    auto L = [&](int i) {
      if constexpr (!(ABL & 1)) d[i] = __builtin_bit_cast(dvis_f4, __builtin_amdgcn_raw_buffer_load_b128(rx, rowq[i], so_load, 0));
    };
    auto item = [&](int m) {   // m = 0..31, after MFMA pair m
      if constexpr (ABL & 2) return;
      if (m < 4) S(m, 0);
      else if (m < 6) T(m - 4);
      else if (m < 10) S(m - 6, 1);
      else if (m < 12) T(m - 8);
      else if (m < 16) { V(m - 12); L(m - 12); }
    };
    b[0][0] = vr[0];
    b[0][1] = vr[(4 * kRow) / 4];
#pragma unroll
    for (int x8 = 0; x8 < 8; ++x8)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int m = (x8 * 2 + s) * 2 + h2;
          if constexpr (!(ABL & 8)) {
            const float av = u[x8 >> 1][(x8 & 1) * 2 + s];
#pragma unroll
            for (int tb = 2 * h2; tb < 2 * h2 + 2; ++tb)
              acc[x8][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[x8 & 1][s][tb], acc[x8][tb], 0, 0, 0);
          }
          if (x8 + 1 < 8 && s == 1) b[(x8 + 1) & 1][h2] = vr[((x8 + 1) * kPos + h2 * 4 * kRow) / 4];   // next position's operands
          item(m);
          fence();
        }
  };
  // V = B^T d B with B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1], position xi = 4 i + jj, to row (xi, channel wv).
  // nch is even (C % 16 == 0): stages in pairs, even ones in s0 / ua, odd ones in s1 / ub, straight-line code (run-time
  // branches inside make hipcc's s_waitcnt insertion assume the fewest loads in flight on any path: it waited for loads it
  // had just issued).  The last pair is not special: it re-requests the last stage (clamped index) and transforms it into a
  // buffer nobody reads any more — 3 % more loads at C = 256; a peeled copy in front of the epilogue made hipcc spill 300-600
  // registers.
  // Patches are requested TWO stages ahead (two register sets): one stage (~1.5 us) is about the HBM latency under load, and
  // the input streams from HBM — with one set the wave waited for its patch at the head of every stage (1.5 of 8.4 ms).
  // (The prologue requests in the order the loop's back edge leaves the loads in — older patch, younger patch, operands: the
  // s_waitcnt counts at the loop head are the merge of both ways in.)
  dvis_f4 d2[4] = {};
  load_d(1, d);
  fence();
  load_d(min(2, nch - 1), d2);
  fence();
  load_u(1, ub);
  fence();
#pragma unroll 1
  for (int ch = 0; ch < nch; ch += 2) {
    const int c2 = min(ch + 2, nch - 1), c3 = min(ch + 3, nch - 1), c4 = min(ch + 4, nch - 1);
    __syncthreads();   // s0 holds stage ch; nobody reads s1 (stage ch - 1) any more
    stage(s0, ua, s1, d, c3);    // ... transforms stage ch + 1 (in d) into s1, then requests the patch of stage ch + 3 into d
    load_u(c2, ua);
    fence();
    __syncthreads();   // s1 holds stage ch + 1; nobody reads s0 any more
    stage(s1, ub, s0, d2, c4);   // ... stage ch + 2 (in d2) into s0, the patch of stage ch + 4 into d2
    load_u(c3, ub);
    fence();
  }

  // ---- output transform.  Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  Rows first, over THIS half's two rows of M (half 0:
  // i = 0, 1; half 1: i = 2, 3), then columns: a partial 2x2 block yp[a][b] per (channel row r, tile block tb).
  float yp[4][4][4];   // [tb][r][2 a + b]
#pragma unroll
  for (int tb = 0; tb < 4; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float R0[4], R1[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float m_lo = acc[jj][tb][r], m_hi = acc[4 + jj][tb][r];
        R0[jj] = half ? m_lo : m_lo + m_hi;
        R1[jj] = half ? -m_lo - m_hi : m_hi;
      }
      yp[tb][r][0] = R0[0] + R0[1] + R0[2];
      yp[tb][r][1] = R0[1] - R0[2] - R0[3];
      yp[tb][r][2] = R1[0] + R1[1] + R1[2];
      yp[tb][r][3] = R1[1] - R1[2] - R1[3];
    }
  // half h stores accumulator tiles tb = 2 h, 2 h + 1 (tiles 4 j + tb); the partials of the other two go to the partner wave
  // through LDS
  __syncthreads();   // the stages are dead
  {
    float *ex = lds + (((1 - half) * 4 + kb16) * 2) * 16 * 64 + lane;   // [dst half][kb16][tb & 1][r][ab][lane]
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ab = 0; ab < 4; ++ab) ex[((t2 * 4 + r) * 4 + ab) * 64] = half ? yp[t2][r][ab] : yp[2 + t2][r][ab];
  }
  __syncthreads();
  const float *ex = lds + ((half * 4 + kb16) * 2) * 16 * 64 + lane;
  const int k0 = kb * kKw + kb16 * 16 + 4 * g;
  float bias4[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = a.bias ? a.bias[k0 + r] : 0.f;
  // this wave's two tiles of the lane: 4 j + 2 half and the next one = 4 adjacent output columns (2 rows)
  float o[2][4][4];   // [t2][r][2 a + b]
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ab = 0; ab < 4; ++ab) {
        const float mine = half ? yp[2 + t2][r][ab] : yp[t2][r][ab];
        const float other = ex[((t2 * 4 + r) * 4 + ab) * 64];
        const float v = (half ? other + mine : mine + other) + bias4[r];   // (half 0's partial) + (half 1's partial)
        o[t2][r][ab] = a.relu ? fmaxf(v, 0.f) : v;
      }
  const long long pa = p0 + 4 * j + 2 * half;
  if (((a.TX & 1) | (a.W & 3)) == 0) {   // the pair lies in one tile row, 16-byte aligned: one float4 per output row
    if (pa < a.tiles) {
      const int n = (int)(pa / per_img), rr = (int)(pa - (long long)n * per_img);
      const int ty = rr / a.TX, tx = rr - ty * a.TX;
      const bool y1 = 2 * ty + 1 < a.H;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *yrow = a.y + (((long long)n * a.K + k0 + r) * a.H + 2 * ty) * a.W + 2 * tx;
        *reinterpret_cast<float4 *>(yrow) = make_float4(o[0][r][0], o[0][r][1], o[1][r][0], o[1][r][1]);
        if (y1) *reinterpret_cast<float4 *>(yrow + a.W) = make_float4(o[0][r][2], o[0][r][3], o[1][r][2], o[1][r][3]);
      }
    }
  } else {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const long long p = pa + t2;
      if (p >= a.tiles) continue;
      const int n = (int)(p / per_img), rr = (int)(p - (long long)n * per_img);
      const int ty = rr / a.TX, tx = rr - ty * a.TX;
      const bool x1 = 2 * tx + 1 < a.W, y1 = 2 * ty + 1 < a.H;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *yrow = a.y + (((long long)n * a.K + k0 + r) * a.H + 2 * ty) * a.W + 2 * tx;
        yrow[0] = o[t2][r][0];
        if (x1) yrow[1] = o[t2][r][1];
        if (y1) {
          yrow[a.W] = o[t2][r][2];
          if (x1) yrow[a.W + 1] = o[t2][r][3];
        }
      }
    }
  }
}

// U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], packed for the kernel's A-operand loads:
//   uf[kb16][stage][half][q][lane = 16 g + i][e = 2 xo + s] = U_xi[k = 16 kb16 + i][c = 8 stage + 4 s + g],  xi = 8 half + 2 q + xo
__global__ void winograd_pack_kernel(const float *__restrict__ w, float *__restrict__ uf, int K, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * C) return;
  const int k = idx / C, c = idx - k * C;
  double gk[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) gk[i][jj] = (double)w[(long long)idx * 9 + i * 3 + jj];
  double t[4][3];
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    t[0][jj] = gk[0][jj];
    t[1][jj] = 0.5 * (gk[0][jj] + gk[1][jj] + gk[2][jj]);
    t[2][jj] = 0.5 * (gk[0][jj] - gk[1][jj] + gk[2][jj]);
    t[3][jj] = gk[2][jj];
  }
  const int kb16 = k >> 4, i16 = k & 15, st = c >> 3, s = (c >> 2) & 1, g = c & 3, nch = C / kCc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double u4[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int xi = 4 * i + jj, half = xi >> 3, q = (xi >> 1) & 3, xo = xi & 1;
      const long long o = ((((long long)(kb16 * nch + st) * 2 + half) * 4 + q) * 64 + 16 * g + i16) * 4 + 2 * xo + s;
      uf[o] = (float)u4[jj];
    }
  }
}

}  // namespace

DVIS_EXPORT int dvis_conv3x3_winograd_supported(int C, int K, int H, int W) {
  if (C <= 0 || K <= 0 || H <= 0 || W < 4 || (W & 1) || C % 16 != 0 || K % kKw != 0) return 0;   // (W even: 16-byte patch rows)
  const long long tiles_per_img = (long long)((H + 1) / 2) * ((W + 1) / 2);
  if (tiles_per_img < kTiles) return 0;
  if (2ll * C * H * W * 4 >= (1ll << 31) || 16ll * K * C * 4 >= (1ll << 31)) return 0;
  return 1;
}

DVIS_EXPORT int dvis_conv3x3_winograd_pack(const float *w, float *uf, int K, int C, void *stream) {
  DVIS_REQUIRE(w && uf && K > 0 && C > 0 && C % 16 == 0 && K % kKw == 0, "conv3x3_winograd_pack: K %% 64 == 0 and C %% 16 == 0");
  const int n = K * C;
  hipLaunchKernelGGL(winograd_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, uf, K, C);
  return dvis_check_launch("dvis_conv3x3_winograd_pack");
}

DVIS_EXPORT int dvis_conv3x3_winograd(const float *x, const float *uf, const float *bias, float *y, int N, int C, int K, int H,
                                      int W, int relu, void *stream) {
  DVIS_REQUIRE(N >= 0, "conv3x3_winograd: bad batch");
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(x && uf && y, "conv3x3_winograd: null pointer");
  DVIS_REQUIRE(dvis_conv3x3_winograd_supported(C, K, H, W),
               "conv3x3_winograd: unsupported shape C=%d K=%d H=%d W=%d (dvis_conv3x3_winograd_supported)", C, K, H, W);
  DVIS_REQUIRE((((uintptr_t)x | (uintptr_t)uf | (uintptr_t)y) & 15) == 0, "conv3x3_winograd: 16-byte aligned tensors");
  WinoArgs a;
  a.x = x, a.uf = uf, a.bias = bias, a.y = y;
  a.N = N, a.C = C, a.K = K, a.H = H, a.W = W, a.relu = relu;
  a.TY = (H + 1) / 2, a.TX = (W + 1) / 2;
  a.tiles = (long long)N * a.TY * a.TX;
  const long long nsp = (a.tiles + kTiles - 1) / kTiles;
  DVIS_REQUIRE(nsp * (K / kKw) + 8 * (K / kKw) < (1ll << 31), "conv3x3_winograd: grid too large");
  a.nsp = (int)nsp;
  const size_t lds_bytes = 2 * kStage * sizeof(float);
  const unsigned grid = (unsigned)(((nsp + 7) / 8) * 8 * (K / kKw));
#ifdef DVIS_WINO_ABLATION   // development builds only (tools/exp/wino_abl.sh): the kernel with parts switched off — WRONG results
  static const int abl = getenv("DVIS_WINO_ABL") ? atoi(getenv("DVIS_WINO_ABL")) : 0;
  switch (abl) {
#define DVIS_WINO_CASE(V)                                                                                          \
  case V:                                                                                                          \
    hipLaunchKernelGGL(winograd_f2x3_kernel<V>, dim3(grid), dim3(512), lds_bytes, (hipStream_t)stream, a);          \
    return dvis_check_launch("dvis_conv3x3_winograd");
    DVIS_WINO_CASE(1) DVIS_WINO_CASE(2) DVIS_WINO_CASE(3) DVIS_WINO_CASE(4) DVIS_WINO_CASE(7) DVIS_WINO_CASE(8) DVIS_WINO_CASE(11)
#undef DVIS_WINO_CASE
  default:
    break;
  }
#endif
  hipLaunchKernelGGL(winograd_f2x3_kernel<0>, dim3(grid), dim3(512), lds_bytes, (hipStream_t)stream, a);
  return dvis_check_launch("dvis_conv3x3_winograd");
}
