// Small fused memory-bound passes around the library GEMMs / convolutions — gfx950.
//
//   dvis_add_layernorm   out = LayerNorm(x + res) * gamma + beta      (post-norm residual blocks of every transformer
//                        layer on the path: msdeformattn.py:125-131, video_mask2former_transformer_decoder.py:47-50,
//                        108-111, 166-170, tracker.py:51-53).  torch runs add and layer_norm as two kernels (3 + 2
//                        passes over the tensor); this is one pass (2 reads, 1 write) with the row kept in registers.
//   dvis_upsample_add    out = lateral + bilinear_upsample(top)        (FPN top-down step, msdeformattn.py:347;
//                        F.interpolate(align_corners=False) to the lateral's size + add: 2 kernels and a 1.8 GB
//                        intermediate per clip in torch).
//   dvis_bias_act        x = act(x + bias[c] (+ residual)) in place on NCHW planes: the folded-FrozenBN bias, the bottleneck
//                        shortcut add and the ReLU after a MIOpen convolution (3 torch kernels -> 1 pass).
// All are HBM-bound; float4 accesses, one wave per row (layernorm) / one thread per 4 output pixels (upsample).
#include <math.h>

#include "dvis_common.h"

namespace {

// One wave per row, C = 4 * 64 * V floats held in registers; two-pass (mean, then centred variance) like torch.
template <int V>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ x, const float *__restrict__ res,
                                                            int64_t res_row_stride, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ out,
                                                            size_t rows, int C, float eps,
                                                            const float *__restrict__ pos, size_t pos_rows,
                                                            float *__restrict__ out_pos) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float4 *xr = reinterpret_cast<const float4 *>(x + row * C);
  const float4 *rr = res ? reinterpret_cast<const float4 *>(res + row * res_row_stride) : nullptr;
  float4 v[V];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 * 4 < C) {
      v[i] = xr[c4];
      if (rr) {
        const float4 r = rr[c4];
        v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w;
      }
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 * 4 < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (c * c + d * d);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)C + eps);
  float4 *orow = reinterpret_cast<float4 *>(out + row * C);
  const float4 *g4 = reinterpret_cast<const float4 *>(gamma);
  const float4 *b4 = reinterpret_cast<const float4 *>(beta);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 * 4 < C) {
      const float4 g = g4[c4], b = b4[c4];
      const float4 o = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                   (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
      orow[c4] = o;
      if (pos) {   // second output: out + pos[row mod pos_rows] — the next layer's query (src + pos), never a pass of its own
        const float4 p = reinterpret_cast<const float4 *>(pos + (row % pos_rows) * C)[c4];
        reinterpret_cast<float4 *>(out_pos + row * C)[c4] = make_float4(o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w);
      }
    }
  }
}

// torch upsample_bilinear2d (align_corners=False) taps:  src = max(scale * (dst + .5) - .5, 0)
// X2 (H == 2h, W == 2w — every FPN step of the pixel decoder): the 4 outputs of a thread read only the 4 columns
// 2xq-1 .. 2xq+2 of two source rows, fetched as one float2 + two scalars per row (6 loads instead of 16; the kernel was
// load-issue bound at 2.8 TB/s).  Weights and tap order are the generic formula's, so both paths give the same bits.
// lat_scale / lat_shift (per plane, may be null): the lateral operand is read as lateral * scale + shift — the
// GroupNorm of the lateral convolution applied on the fly (dvis_group_norm_affine), one pass over the map saved.
template <bool X2>
__global__ __launch_bounds__(256) void upsample_add_kernel(const float *__restrict__ lateral, const float *__restrict__ top,
                                                           float *__restrict__ out, int planes, int H, int W, int h, int w,
                                                           const float *__restrict__ lat_scale,
                                                           const float *__restrict__ lat_shift) {
  const unsigned W4 = (unsigned)W / 4u;
  const unsigned total = (unsigned)planes * (unsigned)H * W4;   // < 2^32 (host check)
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned r = idx / W4;
    const int xq = (int)(idx - r * W4);
    const unsigned pl = r / (unsigned)H;
    const int y = (int)(r - pl * (unsigned)H);
    float fy = sy * ((float)y + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = min((int)fy, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
    const float *t0 = top + ((size_t)pl * h + y0) * (size_t)w, *t1 = top + ((size_t)pl * h + y1) * (size_t)w;
    const size_t o = ((size_t)pl * H + y) * (size_t)W + 4 * xq;
    float4 lat = *reinterpret_cast<const float4 *>(lateral + o);
    if (lat_scale) {
      const float a = lat_scale[pl], b = lat_shift[pl];
      lat.x = lat.x * a + b; lat.y = lat.y * a + b; lat.z = lat.z * a + b; lat.w = lat.w * a + b;
    }
    float c0[4], c1[4];     // X2: source columns 2xq-1 (clamped), 2xq, 2xq+1, 2xq+2 (clamped) of rows y0 / y1
    if (X2) {
      const int xm = max(2 * xq - 1, 0), xp = min(2 * xq + 2, w - 1);
      const float2 m0 = *reinterpret_cast<const float2 *>(t0 + 2 * xq), m1 = *reinterpret_cast<const float2 *>(t1 + 2 * xq);
      c0[0] = t0[xm]; c0[1] = m0.x; c0[2] = m0.y; c0[3] = t0[xp];
      c1[0] = t1[xm]; c1[1] = m1.x; c1[2] = m1.y; c1[3] = t1[xp];
    }
    float res[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float fx = sx * ((float)(4 * xq + k) + 0.5f) - 0.5f;
      fx = fx < 0.f ? 0.f : fx;
      const int x0 = min((int)fx, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
      if (X2) {
        // x0 = 2xq-1, 2xq, 2xq, 2xq+1 for k = 0..3 (x0 = 0 with lx1 = 0 at the left border: column 0 either way)
        const int i0 = (k + 1) / 2, i1 = i0 + 1;
        res[k] = ly0 * (lx0 * c0[i0] + lx1 * c0[i1]) + ly1 * (lx0 * c1[i0] + lx1 * c1[i1]);
      } else {
        res[k] = ly0 * (lx0 * t0[x0] + lx1 * t0[x1]) + ly1 * (lx0 * t1[x0] + lx1 * t1[x1]);
      }
    }
    *reinterpret_cast<float4 *>(out + o) = make_float4(lat.x + res[0], lat.y + res[1], lat.z + res[2], lat.w + res[3]);
  }
}

// any W (rows not a multiple of 4 floats): one output per thread, the generic formula of the kernel above
__global__ __launch_bounds__(256) void upsample_add_scalar_kernel(const float *__restrict__ lateral, const float *__restrict__ top,
                                                                  float *__restrict__ out, long long total, int H, int W, int h, int w,
                                                                  const float *__restrict__ lat_scale,
                                                                  const float *__restrict__ lat_shift) {
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int x = (int)(idx % W);
    const long long r = idx / W;
    const int y = (int)(r % H);
    const long long pl = r / H;
    float fy = sy * ((float)y + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = min((int)fy, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
    float fx = sx * ((float)x + 0.5f) - 0.5f;
    fx = fx < 0.f ? 0.f : fx;
    const int x0 = min((int)fx, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
    const float *t0 = top + ((size_t)pl * h + y0) * (size_t)w, *t1 = top + ((size_t)pl * h + y1) * (size_t)w;
    float lat = lateral[idx];
    if (lat_scale) lat = lat * lat_scale[pl] + lat_shift[pl];
    out[idx] = lat + (ly0 * (lx0 * t0[x0] + lx1 * t0[x1]) + ly1 * (lx0 * t1[x0] + lx1 * t1[x1]));
  }
}

// out = relu(max_pool2d(x, 3, stride 2, padding 1) + bias[c]) — the ResNet stem after its convolution.  x -> relu(x + b)
// is monotonic in fp32, so this equals max_pool2d(relu(x + bias)) bit for bit; the stem output (the largest activation
// of the model) is read once and never written back.  A thread makes 4 outputs of one row from 9 input columns x 3 rows.
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(const float *__restrict__ x, const float *__restrict__ bias,
                                                                float *__restrict__ out, int planes, int C, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const unsigned Wq = (unsigned)Wo / 4u;
  const unsigned total = (unsigned)planes * (unsigned)Ho * Wq;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned r = idx / Wq;
    const int xq = (int)(idx - r * Wq);
    const unsigned pl = r / (unsigned)Ho;
    const int oy = (int)(r - pl * (unsigned)Ho);
    const float *xp = x + (size_t)pl * H * W;
    float m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) m[i] = -INFINITY;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int y = 2 * oy + dy;
      if (y < 0) continue;                          // padding row (y <= H - 1 always: H is even)
      const float *row = xp + (size_t)y * W + 8 * xq;
      const float4 a = *reinterpret_cast<const float4 *>(row), b = *reinterpret_cast<const float4 *>(row + 4);
      const float l = xq > 0 ? row[-1] : -INFINITY;   // padding column
      m[0] = fmaxf(m[0], l);
      m[1] = fmaxf(m[1], a.x); m[2] = fmaxf(m[2], a.y); m[3] = fmaxf(m[3], a.z); m[4] = fmaxf(m[4], a.w);
      m[5] = fmaxf(m[5], b.x); m[6] = fmaxf(m[6], b.y); m[7] = fmaxf(m[7], b.z); m[8] = fmaxf(m[8], b.w);
    }
    const float bb = bias ? bias[pl % (unsigned)C] : 0.f;
    float4 o;
    o.x = fmaxf(fmaxf(fmaxf(m[0], m[1]), m[2]) + bb, 0.f);
    o.y = fmaxf(fmaxf(fmaxf(m[2], m[3]), m[4]) + bb, 0.f);
    o.z = fmaxf(fmaxf(fmaxf(m[4], m[5]), m[6]) + bb, 0.f);
    o.w = fmaxf(fmaxf(fmaxf(m[6], m[7]), m[8]) + bb, 0.f);
    *reinterpret_cast<float4 *>(out + ((size_t)pl * Ho + oy) * Wo + 4 * xq) = o;
  }
}

// x[(n*C + c)*HW + i] = relu?(x + bias[c] + res); grid.x = planes * chunks_per_plane
__global__ __launch_bounds__(256) void bias_act_kernel(float *__restrict__ x, const float *__restrict__ bias,
                                                       const float *__restrict__ res, int C, int HW4, int chunks,
                                                       int relu) {
  const unsigned plane = blockIdx.x / chunks;     // n * C + c
  const int chunk = blockIdx.x - plane * chunks;
  const float b = bias ? bias[plane % C] : 0.f;
  float4 *xp = reinterpret_cast<float4 *>(x) + (size_t)plane * HW4;
  const float4 *rp = res ? reinterpret_cast<const float4 *>(res) + (size_t)plane * HW4 : nullptr;
  for (int i = chunk * 256 + threadIdx.x; i < HW4; i += chunks * 256) {
    float4 v = xp[i];
    v.x += b; v.y += b; v.z += b; v.w += b;
    if (rp) {
      const float4 r = rp[i];
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    xp[i] = v;
  }
}

// the same per element for planes that are not a multiple of 4 floats (odd x odd stride-32 maps, e.g. 15 x 27 from a 480 x 854
// frame): planes are then not 16-byte aligned, so one float per thread; same operations in the same order -> same bits
__global__ __launch_bounds__(256) void bias_act_scalar_kernel(float *__restrict__ x, const float *__restrict__ bias,
                                                              const float *__restrict__ res, int C, long long HW, long long total,
                                                              int relu) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    float v = x[i] + (bias ? bias[(i / HW) % C] : 0.f);
    if (res) v += res[i];
    x[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

// out[n][row0 + p][c] = x[n][c][p]: an (N, C, HW) map laid down as HW token rows of a wider (N, S, C) token matrix —
// the flatten(2).transpose(1, 2) + cat of MSDeformAttnTransformerEncoderOnly.forward (msdeformattn.py:64-79) without
// torch's strided-copy kernel (0.6 TB/s).  64 x 64 tiles through LDS: reads run along p, writes along c.
__global__ __launch_bounds__(256) void nchw_to_tokens_kernel(const float *__restrict__ x, float *__restrict__ out, int C,
                                                             int HW, int64_t S, int64_t row0,
                                                             const float *__restrict__ scale, const float *__restrict__ shift,
                                                             const float *__restrict__ pos, float *__restrict__ out_pos) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float *xb = x + (size_t)n * C * HW;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, p = p0 + tx;
    float v = (c < C && p < HW) ? xb[(size_t)c * HW + p] : 0.f;
    if (scale && c < C) v = v * scale[(size_t)n * C + c] + shift[(size_t)n * C + c];   // the map's GroupNorm, per plane
    tile[ty + 4 * i][tx] = v;
  }
  __syncthreads();
  float *ob = out + ((size_t)n * S + row0) * C;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = p0 + ty + 4 * i, c = c0 + tx;
    if (p < HW && c < C) {
      const float v = tile[tx][ty + 4 * i];
      ob[(size_t)p * C + c] = v;
      if (pos)   // second output: tokens + position embedding (the first encoder layer's query)
        out_pos[((size_t)n * S + row0 + p) * C + c] = v + pos[((size_t)row0 + p) * C + c];
    }
  }
}

// The way back: out[n][c][p] = tok[n][row0 + p][c] — one level of the encoder's token matrix as the (N, C, h, w) map the FPN's
// top-down path reads (msdeformattn.py:333-339 `y[:, :, ...].transpose(1, 2).view(bs, -1, h, w)`; the reference leaves it a
// strided view and pays in the consumer).  64 x 64 tiles through LDS: reads run along c, writes along p.
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const float *__restrict__ tok, float *__restrict__ out, int C, int HW,
                                                             int64_t S, int64_t row0) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float *tb = tok + ((size_t)n * S + row0) * C;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int p = p0 + ty + 4 * i, c = c0 + tx;
    tile[ty + 4 * i][tx] = (p < HW && c < C) ? tb[(size_t)p * C + c] : 0.f;
  }
  __syncthreads();
  float *ob = out + (size_t)n * C * HW;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, p = p0 + tx;
    if (c < C && p < HW) ob[(size_t)c * HW + p] = tile[tx][ty + 4 * i];
  }
}

// Input normalisation + padding of a clip in one pass: out[p][y][x] = y < H && x < W ? (float(in[p][y][x]) - mean[p % C]) / std[p % C] : 0
// (dvis_Plus/meta_architecture.py:1310-1311: `(x - pixel_mean) / pixel_std`, then ImageList.from_tensors pads with zeros to the
// size divisibility).  Same fp32 operations in the same order as the torch expression (IEEE subtract, IEEE divide): same bits,
// one read of the uint8 / float frames and one write of the padded tensor instead of conversion + sub + div + fill + copy.
template <typename T>
__global__ __launch_bounds__(256) void normalize_pad_kernel(const T *__restrict__ in, float *__restrict__ out, int C, int H, int W,
                                                            int Hp, int Wp, const float *__restrict__ mean,
                                                            const float *__restrict__ stdv) {
  // block = 64 groups of 4 pixels x 4 rows; grid = (groups of a row / 64, rows / 4, planes)
  const int g = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int64_t p = blockIdx.z;
  if (4 * g >= Wp || y >= Hp) return;
  const int c = (int)(p % C);
  const float m = mean[c], sd = stdv[c];
  const T *src = in + (p * H + y) * (int64_t)W;
  float *dst = out + (p * Hp + y) * (int64_t)Wp;
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = 4 * g + k;
    v[k] = (y < H && x < W) ? __fdiv_rn(__fsub_rn((float)src[x], m), sd) : 0.f;
  }
  if ((Wp & 3) == 0) {
    *reinterpret_cast<dvis_f4 *>(dst + 4 * g) = dvis_f4{v[0], v[1], v[2], v[3]};
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * g + k < Wp) dst[4 * g + k] = v[k];
  }
}

// ---- ViT-Adapter stride-4 output (adapter.py:341-356: c1 = up(c2) + c1; c1 += interpolate(x1, 4); f1 = norm1(c1)) --------------
// `up` is a 2 x 2 / stride-2 transposed convolution = a GEMM over the stride-8 tokens with 4 C output features (dy, dx, co); its
// result g arrives TOKEN-major.  This kernel is the pixel shuffle back to NCHW with everything that follows folded in:
//     out[b, co, 2y+dy, 2x+dx] = g[(b, y, x), (dy, dx, co)] + scale[co] * (c1[b, co, 2y+dy, 2x+dx] + up4(x1)[...]) + shift[co]
// (eval BatchNorm scale folded into the GEMM's weights by the caller; shift = scale * up.bias + BN shift).  One pass over the
// 7.2 GB (30 frames, 1024 channels, 184 x 320) instead of transposed-conv output + two adds + upsample + BN + a layout copy.
// Tile: 32 tokens of one stride-8 row x 64 channels: g through LDS (read along co, written along x), the two x1 source rows the
// tile's two output rows interpolate between (18 columns x 64 channels) staged the same way.
__global__ __launch_bounds__(256) void adapter_res2_kernel(const float *__restrict__ g, const float *__restrict__ c1,
                                                           const float *__restrict__ x1, const float *__restrict__ scale,
                                                           const float *__restrict__ shift, float *__restrict__ out, int C, int h8,
                                                           int w8) {
  constexpr int TS = 65;                               // token stride in LDS (floats)
  __shared__ float gs[4 * 32 * TS];                    // [sub][token][co]
  __shared__ float xs[2 * 18 * TS];                    // [row][col][co]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tx = blockIdx.x * 32, co0 = blockIdx.y * 64;
  const int b = blockIdx.z / h8, y = blockIdx.z - b * h8;
  const int H4 = 2 * h8, W4 = 2 * w8, h16 = h8 / 2, w16 = w8 / 2;
  const int ntok = min(32, w8 - tx);
  // g: rows of 4 C floats per token; 16 lanes read one (token, sub)'s 64 channels as float4, a wave four such rows per instruction
  {
    const int cl = lane & 15, rr = lane >> 4;
    for (int r = wv * 4 + rr; r < 128; r += 16) {
      const int tok = r >> 2, sub = r & 3;
      if (tok < ntok) {
        const float4 v = *reinterpret_cast<const float4 *>(g + (((size_t)b * h8 + y) * w8 + tx + tok) * (size_t)(4 * C) + (size_t)sub * C + co0 + 4 * cl);
        float *d = &gs[(sub * 32 + tok) * TS + 4 * cl];
        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
      }
    }
  }
  // x1 source rows / columns of this tile (align_corners = False, factor 4: src = (dst + 0.5) / 4 - 0.5, clamped at 0)
  const int ys0 = max((y >> 1) - ((y & 1) ? 0 : 1), 0), ys1 = min(((y >> 1) - ((y & 1) ? 0 : 1)) + 1, h16 - 1);
  const int xc0 = (tx >> 1) - 1;                       // first staged column (may be -1: clamped on load)
  if (x1 != nullptr) {
    for (int r = wv; r < 36; r += 4) {
      const int row = r / 18, col = r - row * 18;
      const int ysrc = row ? ys1 : ys0, xsrc = min(max(xc0 + col, 0), w16 - 1);
      xs[(row * 18 + col) * TS + lane] = x1[(((size_t)b * h16 + ysrc) * w16 + xsrc) * (size_t)C + co0 + lane];
    }
  }
  __syncthreads();
  // output: a lane writes 4 consecutive columns (16 bytes) of one (channel, dy) row, a wave four rows per instruction
  const int xl = lane & 15, rr = lane >> 4;
  const bool x_ok = 4 * xl < 2 * ntok;                 // (w8 even -> ntok even: whole float4s)
  int cA[4], cB[4];
  float lx0[4], lx1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float fx = ((float)(2 * tx + 4 * xl + e) + 0.5f) * 0.25f - 0.5f;
    fx = fx < 0.f ? 0.f : fx;
    const int xi0 = min((int)fx, w16 - 1), xi1 = min(xi0 + 1, w16 - 1);
    lx1[e] = fx - (float)xi0, lx0[e] = 1.f - lx1[e];
    cA[e] = min(max(xi0 - xc0, 0), 17), cB[e] = min(max(xi1 - xc0, 0), 17);      // staged columns (lanes past the row's end: clamped, unused)
  }
  for (int R = wv * 4 + rr; R < 128; R += 16) {
    const int c = R >> 1, dy = R & 1, co = co0 + c;
    const int Y = 2 * y + dy;
    float up[4] = {0.f, 0.f, 0.f, 0.f};
    if (x1 != nullptr) {
      float fy = ((float)Y + 0.5f) * 0.25f - 0.5f;
      fy = fy < 0.f ? 0.f : fy;
      const int yi0 = min((int)fy, h16 - 1);
      const float ly1 = fy - (float)yi0, ly0 = 1.f - ly1;
      const int rA = yi0 == ys0 ? 0 : 1, rB = (min(yi0 + 1, h16 - 1)) == ys0 ? 0 : 1;
      // the order of F.interpolate's bilinear kernel: ly0 (lx0 v00 + lx1 v01) + ly1 (lx0 v10 + lx1 v11)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        up[e] = ly0 * (lx0[e] * xs[(rA * 18 + cA[e]) * TS + c] + lx1[e] * xs[(rA * 18 + cB[e]) * TS + c]) +
                ly1 * (lx0[e] * xs[(rB * 18 + cA[e]) * TS + c] + lx1[e] * xs[(rB * 18 + cB[e]) * TS + c]);
    }
    if (x_ok) {
      const size_t o = (((size_t)b * C + co) * H4 + Y) * (size_t)W4 + 2 * tx + 4 * xl;
      const float4 cv = *reinterpret_cast<const float4 *>(c1 + o);
      const float sc = scale[co], sh = shift[co];
      const float *gq = &gs[(dy * 2 * 32 + 2 * xl) * TS + c];          // (sub = 2 dy + dx, token 2 xl + t): columns 4 xl + 2 t + dx
      float4 v;
      v.x = gq[0] + sc * (cv.x + up[0]) + sh;
      v.y = gq[32 * TS] + sc * (cv.y + up[1]) + sh;
      v.z = gq[TS] + sc * (cv.z + up[2]) + sh;
      v.w = gq[33 * TS] + sc * (cv.w + up[3]) + sh;
      *reinterpret_cast<float4 *>(out + o) = v;
    }
  }
}

// ---- depthwise 3 x 3 convolution (+ bias, + GELU) on TOKEN-major maps: the ConvFFN of the ViT-Adapter extractors
// (adapter_modules.py DWConv: three pyramid levels stored one after the other in a (B, N, C) token tensor; the reference
// transposes each level to NCHW, runs a grouped Conv2d, transposes back and concatenates).  Channels are the contiguous axis here,
// so a lane keeps 4 channels' 9 taps in registers and walks 4 neighbouring tokens of a row: 18 loads of 16 bytes per 4 outputs.
__global__ __launch_bounds__(256) void dwconv3x3_tokens_kernel(const float *__restrict__ x, float *__restrict__ out, int64_t batch_stride,
                                                               int B, int h, int w, int C, const float *__restrict__ wt,
                                                               const float *__restrict__ bias, int gelu) {
  const int c4 = (blockIdx.y * 64 + (threadIdx.x & 63)) * 4;
  if (c4 >= C) return;
  const int wq = (w + 3) / 4;
  const int64_t strip = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b, y, x-quad)
  const int xq = (int)(strip % wq);
  const int64_t r = strip / wq;
  const int y = (int)(r % h);
  const int64_t b = r / h;
  if (b >= B) return;
  float4 k[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) k[t] = float4{wt[(c4 + 0) * 9 + t], wt[(c4 + 1) * 9 + t], wt[(c4 + 2) * 9 + t], wt[(c4 + 3) * 9 + t]};
  const float4 bv = bias ? *reinterpret_cast<const float4 *>(bias + c4) : float4{0.f, 0.f, 0.f, 0.f};
  const float *xb = x + b * batch_stride + c4;
  float4 acc[4] = {bv, bv, bv, bv};
  const int x0 = 4 * xq;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = y + dy - 1;
    if (yy < 0 || yy >= h) continue;
    float4 col[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int xx = x0 + i - 1;
      col[i] = xx >= 0 && xx < w ? *reinterpret_cast<const float4 *>(xb + ((int64_t)yy * w + xx) * C) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float4 kk = k[dy * 3 + dx], v = col[o + dx];
        acc[o].x = __builtin_fmaf(kk.x, v.x, acc[o].x), acc[o].y = __builtin_fmaf(kk.y, v.y, acc[o].y);
        acc[o].z = __builtin_fmaf(kk.z, v.z, acc[o].z), acc[o].w = __builtin_fmaf(kk.w, v.w, acc[o].w);
      }
  }
  float *ob = out + b * batch_stride + c4;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    if (x0 + o >= w) break;
    float4 v = acc[o];
    if (gelu) {
      v.x = 0.5f * v.x * (1.f + erff(v.x * 0.70710678118654752440f)), v.y = 0.5f * v.y * (1.f + erff(v.y * 0.70710678118654752440f));
      v.z = 0.5f * v.z * (1.f + erff(v.z * 0.70710678118654752440f)), v.w = 0.5f * v.w * (1.f + erff(v.w * 0.70710678118654752440f));
    }
    *reinterpret_cast<float4 *>(ob + ((int64_t)y * w + x0 + o) * C) = v;
  }
}

}  // namespace

DVIS_EXPORT int dvis_normalize_pad(const void *in, int is_u8, float *out, int64_t planes, int C, int H, int W, int Hp, int Wp,
                                   const float *mean, const float *stdv, void *stream) {
  DVIS_REQUIRE(planes >= 0 && C > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W, "normalize_pad: bad sizes");
  if (planes == 0) return DVIS_OK;
  DVIS_REQUIRE(in && out && mean && stdv, "normalize_pad: null pointer");
  DVIS_REQUIRE(planes <= 65535 && (Hp + 3) / 4 <= 65535, "normalize_pad: grid too large");
  const dim3 grid((unsigned)(((Wp + 3) / 4 + 63) / 64), (unsigned)((Hp + 3) / 4), (unsigned)planes);
  if (is_u8)
    hipLaunchKernelGGL(normalize_pad_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char *)in, out, C,
                       H, W, Hp, Wp, mean, stdv);
  else
    hipLaunchKernelGGL(normalize_pad_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)in, out, C, H, W, Hp, Wp,
                       mean, stdv);
  return dvis_check_launch("normalize_pad_kernel");
}

DVIS_EXPORT int dvis_tokens_to_nchw(const float *tok, float *out, int64_t N, int C, int64_t HW, int64_t S, int64_t row0, void *stream) {
  DVIS_REQUIRE(N >= 0 && C > 0 && HW > 0 && S >= HW && row0 >= 0 && row0 + HW <= S, "tokens_to_nchw: bad sizes");
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(tok && out, "tokens_to_nchw: null pointer");
  DVIS_REQUIRE(N <= 65535 && (C + 63) / 64 <= 65535 && HW < (1ll << 31), "tokens_to_nchw: grid too large");
  hipLaunchKernelGGL(tokens_to_nchw_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)N), dim3(256), 0,
                     (hipStream_t)stream, tok, out, C, (int)HW, S, row0);
  return dvis_check_launch("tokens_to_nchw_kernel");
}

static int nchw_to_tokens_launch(const float *x, float *out, int64_t N, int C, int64_t HW, int64_t S, int64_t row0,
                                 const float *scale, const float *shift, const float *pos, float *out_pos, void *stream) {
  DVIS_REQUIRE(N >= 0 && C > 0 && HW > 0 && S >= HW && row0 >= 0 && row0 + HW <= S, "nchw_to_tokens: bad sizes");
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(x && out, "nchw_to_tokens: null pointer");
  DVIS_REQUIRE((scale == nullptr) == (shift == nullptr) && (pos == nullptr) == (out_pos == nullptr),
               "nchw_to_tokens: scale/shift and pos/out_pos come in pairs");
  DVIS_REQUIRE(N <= 65535 && (C + 63) / 64 <= 65535 && HW < (1ll << 31), "nchw_to_tokens: grid too large");
  hipLaunchKernelGGL(nchw_to_tokens_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)N), dim3(256),
                     0, (hipStream_t)stream, x, out, C, (int)HW, S, row0, scale, shift, pos, out_pos);
  return dvis_check_launch("nchw_to_tokens_kernel");
}

DVIS_EXPORT int dvis_nchw_to_tokens(const float *x, float *out, int64_t N, int C, int64_t HW, int64_t S, int64_t row0,
                                    void *stream) {
  return nchw_to_tokens_launch(x, out, N, C, HW, S, row0, nullptr, nullptr, nullptr, nullptr, stream);
}

DVIS_EXPORT int dvis_nchw_to_tokens_affine(const float *x, const float *scale, const float *shift, const float *pos,
                                           float *out, float *out_pos, int64_t N, int C, int64_t HW, int64_t S,
                                           int64_t row0, void *stream) {
  return nchw_to_tokens_launch(x, out, N, C, HW, S, row0, scale, shift, pos, out_pos, stream);
}

DVIS_EXPORT int dvis_bias_act(float *x, const float *bias, const float *res, int64_t planes, int C, int64_t HW, int relu,
                              void *stream) {
  DVIS_REQUIRE(planes >= 0 && C > 0 && HW > 0, "bias_act: bad sizes");
  if (planes == 0) return DVIS_OK;
  DVIS_REQUIRE(x, "bias_act: null pointer");
  if (HW % 4 != 0 || (((uintptr_t)x | (uintptr_t)res) & 15) != 0) {       // odd planes: the scalar form
    const long long total = (long long)planes * HW;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(bias_act_scalar_kernel, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, x, bias, res, C, (long long)HW, total, relu);
    return dvis_check_launch("bias_act_scalar_kernel");
  }
  const int HW4 = (int)(HW / 4);
  int chunks = (HW4 + 1023) / 1024;           // ~4 float4 per thread
  if (chunks < 1) chunks = 1;
  DVIS_REQUIRE(planes * chunks < (1ll << 31), "bias_act: too many planes");
  hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)(planes * chunks)), dim3(256), 0, (hipStream_t)stream, x, bias, res, C,
                     HW4, chunks, relu);
  return dvis_check_launch("bias_act_kernel");
}

static int add_layernorm_launch(const float *x, const float *res, int64_t res_row_stride, const float *gamma,
                                const float *beta, float *out, int64_t rows, int C, float eps, const float *pos,
                                int64_t pos_rows, float *out_pos, void *stream) {
  DVIS_REQUIRE(rows >= 0 && C > 0, "add_layernorm: bad sizes");
  if (rows == 0) return DVIS_OK;
  DVIS_REQUIRE(x && gamma && beta && out, "add_layernorm: null pointer");
  DVIS_REQUIRE(C % 4 == 0 && C <= 1024, "add_layernorm: C must be a multiple of 4 and <= 1024 (got %d)", C);
  DVIS_REQUIRE(res == nullptr || res_row_stride % 4 == 0, "add_layernorm: residual row stride must be a multiple of 4");
  DVIS_REQUIRE(pos == nullptr || (pos_rows > 0 && out_pos != nullptr), "add_layernorm: pos needs pos_rows > 0 and out_pos");
  const uintptr_t al = (uintptr_t)x | (uintptr_t)res | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out |
                       (uintptr_t)pos | (uintptr_t)out_pos;
  DVIS_REQUIRE((al & 15) == 0, "add_layernorm: pointers must be 16-byte aligned");
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const size_t pr = pos ? (size_t)pos_rows : 1;
  if (C <= 256)
    hipLaunchKernelGGL((add_layernorm_kernel<1>), grid, block, 0, st, x, res, res_row_stride, gamma, beta, out, (size_t)rows, C, eps, pos, pr, out_pos);
  else if (C <= 512)
    hipLaunchKernelGGL((add_layernorm_kernel<2>), grid, block, 0, st, x, res, res_row_stride, gamma, beta, out, (size_t)rows, C, eps, pos, pr, out_pos);
  else
    hipLaunchKernelGGL((add_layernorm_kernel<4>), grid, block, 0, st, x, res, res_row_stride, gamma, beta, out, (size_t)rows, C, eps, pos, pr, out_pos);
  return dvis_check_launch("add_layernorm_kernel");
}

DVIS_EXPORT int dvis_add_layernorm(const float *x, const float *res, int64_t res_row_stride, const float *gamma,
                                   const float *beta, float *out, int64_t rows, int C, float eps, void *stream) {
  return add_layernorm_launch(x, res, res_row_stride, gamma, beta, out, rows, C, eps, nullptr, 0, nullptr, stream);
}

DVIS_EXPORT int dvis_add_layernorm_pos(const float *x, const float *res, int64_t res_row_stride, const float *gamma,
                                       const float *beta, float *out, const float *pos, int64_t pos_rows, float *out_pos,
                                       int64_t rows, int C, float eps, void *stream) {
  DVIS_REQUIRE(pos && out_pos, "add_layernorm_pos: null pos / out_pos");
  return add_layernorm_launch(x, res, res_row_stride, gamma, beta, out, rows, C, eps, pos, pos_rows, out_pos, stream);
}

static int upsample_add_launch(const float *lateral, const float *top, float *out, int64_t planes, int H, int W, int h,
                               int w, const float *lat_scale, const float *lat_shift, void *stream) {
  DVIS_REQUIRE(planes >= 0 && H > 0 && W > 0 && h > 0 && w > 0, "upsample_add: bad sizes");
  if (planes == 0) return DVIS_OK;
  DVIS_REQUIRE(lateral && top && out, "upsample_add: null pointer");
  DVIS_REQUIRE((lat_scale == nullptr) == (lat_shift == nullptr), "upsample_add: scale and shift come together");
  DVIS_REQUIRE(planes < (1ll << 31), "upsample_add: too many planes");
  if (W % 4 != 0 || (((uintptr_t)lateral | (uintptr_t)out) & 15) != 0) {
    const long long tot = (long long)planes * H * W;
    long long nb = (tot + 255) / 256;
    if (nb > 256 * 32) nb = 256 * 32;
    hipLaunchKernelGGL(upsample_add_scalar_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, lateral, top, out, tot, H, W,
                       h, w, lat_scale, lat_shift);
    return dvis_check_launch("upsample_add_scalar_kernel");
  }
  const size_t total = (size_t)planes * H * (W / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  DVIS_REQUIRE(total < (1ull << 32), "upsample_add: more than 2^32 output quads");
  const bool x2 = H == 2 * h && W == 2 * w && (w & 1) == 0 && ((uintptr_t)top & 7) == 0;
  if (x2)
    hipLaunchKernelGGL(upsample_add_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lateral, top,
                       out, (int)planes, H, W, h, w, lat_scale, lat_shift);
  else
    hipLaunchKernelGGL(upsample_add_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, lateral, top,
                       out, (int)planes, H, W, h, w, lat_scale, lat_shift);
  return dvis_check_launch("upsample_add_kernel");
}

DVIS_EXPORT int dvis_dwconv3x3_tokens(const float *x, float *out, int64_t batch_stride, int B, int h, int w, int C, const float *weight,
                                      const float *bias, int gelu, void *stream) {
  DVIS_REQUIRE(x && out && weight, "dwconv3x3_tokens: null pointer");
  DVIS_REQUIRE(B >= 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0 && batch_stride % 4 == 0 && batch_stride >= (int64_t)h * w * C,
               "dwconv3x3_tokens: C %% 4 == 0 and a batch stride that is a multiple of 4 floats >= h * w * C are required");
  DVIS_REQUIRE(((uintptr_t)x | (uintptr_t)out | (uintptr_t)(bias ? bias : weight)) % 16 == 0, "dwconv3x3_tokens: 16-byte alignment");
  DVIS_REQUIRE(x != out, "dwconv3x3_tokens: in place is not possible (neighbouring tokens are read)");
  if (B == 0) return DVIS_OK;
  const int64_t strips = (int64_t)h * ((w + 3) / 4);
  DVIS_REQUIRE(strips * B < ((int64_t)1 << 32), "dwconv3x3_tokens: too many tokens");
  hipLaunchKernelGGL(dwconv3x3_tokens_kernel, dim3((unsigned)((strips * B + 3) / 4), (C / 4 + 63) / 64), dim3(256), 0, (hipStream_t)stream, x,
                     out, batch_stride, B, h, w, C, weight, bias, gelu);
  return dvis_check_launch("dwconv3x3_tokens_kernel");
}

DVIS_EXPORT int dvis_adapter_res2(const float *g, const float *c1, const float *x1, const float *scale, const float *shift, float *out,
                                  int B, int C, int h8, int w8, void *stream) {
  DVIS_REQUIRE(g && c1 && scale && shift && out, "adapter_res2: null pointer");
  DVIS_REQUIRE(((uintptr_t)g | (uintptr_t)c1 | (uintptr_t)out) % 16 == 0, "adapter_res2: g / c1 / out must be 16-byte aligned");
  DVIS_REQUIRE(B >= 0 && C > 0 && C % 64 == 0 && h8 > 0 && w8 > 0 && h8 % 2 == 0 && w8 % 2 == 0,
               "adapter_res2: C %% 64 == 0 and an even stride-8 grid are required (C %d, grid %d x %d)", C, h8, w8);
  DVIS_REQUIRE((int64_t)B * h8 <= 65535 && C / 64 <= 65535, "adapter_res2: grid too large (B * rows %lld)", (long long)B * h8);
  if (B == 0) return DVIS_OK;
  hipLaunchKernelGGL(adapter_res2_kernel, dim3((w8 + 31) / 32, C / 64, B * h8), dim3(256), 0, (hipStream_t)stream, g, c1, x1, scale, shift,
                     out, C, h8, w8);
  return dvis_check_launch("adapter_res2_kernel");
}

DVIS_EXPORT int dvis_upsample_add(const float *lateral, const float *top, float *out, int64_t planes, int H, int W, int h,
                                  int w, void *stream) {
  return upsample_add_launch(lateral, top, out, planes, H, W, h, w, nullptr, nullptr, stream);
}

DVIS_EXPORT int dvis_upsample_add_affine(const float *lateral, const float *lat_scale, const float *lat_shift,
                                         const float *top, float *out, int64_t planes, int H, int W, int h, int w,
                                         void *stream) {
  DVIS_REQUIRE(lat_scale && lat_shift, "upsample_add_affine: null scale / shift");
  return upsample_add_launch(lateral, top, out, planes, H, W, h, w, lat_scale, lat_shift, stream);
}

DVIS_EXPORT int dvis_bias_relu_maxpool(const float *x, const float *bias, float *out, int64_t planes, int C, int H, int W,
                                       void *stream) {
  DVIS_REQUIRE(planes >= 0 && C > 0 && H > 0 && W > 0, "bias_relu_maxpool: bad sizes");
  if (planes == 0) return DVIS_OK;
  DVIS_REQUIRE(x && out, "bias_relu_maxpool: null pointer");
  DVIS_REQUIRE(H % 2 == 0 && W % 8 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0,
               "bias_relu_maxpool: needs even H, W %% 8 == 0 and 16-byte aligned x / out (H=%d W=%d)", H, W);
  const size_t total = (size_t)planes * (H / 2) * (W / 8);
  DVIS_REQUIRE(planes < (1ll << 31) && total < (1ull << 32), "bias_relu_maxpool: too many planes / outputs");
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(bias_relu_maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, bias, out,
                     (int)planes, C, H, W);
  return dvis_check_launch("bias_relu_maxpool_kernel");
}

namespace {

// One workgroup per (sample, group): sum and sum of squares of the group's Cg * HW contiguous floats in fp64 (the kernel
// is HBM-bound; fp64 adds are free and make E[x^2] - E[x]^2 safe), then scale = rstd * gamma[c], shift = beta[c] -
// mean * scale for the group's Cg planes.
__global__ __launch_bounds__(1024) void group_norm_affine_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float *__restrict__ scale,
                                                                 float *__restrict__ shift, int C, int G, long long HW,
                                                                 float eps, int vec) {
  __shared__ double s_sum[16], s_sq[16];
  const int Cg = C / G;
  const int n = blockIdx.x / G, g = blockIdx.x - n * G;
  const long long count = (long long)Cg * HW, count4 = count / 4;
  const float4 *xp = reinterpret_cast<const float4 *>(x + ((size_t)n * C + (size_t)g * Cg) * HW);
  double sum = 0.0, sq = 0.0;
  if (vec) {
    for (long long i = threadIdx.x; i < count4; i += 1024) {
      const float4 v = xp[i];
      sum += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      sq += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
  } else {      // a group that is not a multiple of 4 floats (odd x odd maps): its slab is not 16-byte aligned
    const float *xs = reinterpret_cast<const float *>(xp);
    for (long long i = threadIdx.x; i < count; i += 1024) {
      const double v = (double)xs[i];
      sum += v;
      sq += v * v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_down(sum, o);
    sq += __shfl_down(sq, o);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_sum[wv] = sum;
    s_sq[wv] = sq;
  }
  __syncthreads();
  if (threadIdx.x < Cg) {
    double ts = 0.0, tq = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      ts += s_sum[i];
      tq += s_sq[i];
    }
    const double mean = ts / (double)count;
    double var = tq / (double)count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const int c = g * Cg + threadIdx.x;
    const float a = rstd * (gamma ? gamma[c] : 1.f);
    scale[(size_t)n * C + c] = a;
    shift[(size_t)n * C + c] = (beta ? beta[c] : 0.f) - (float)mean * a;
  }
}

// x[plane][i] = relu?(x * scale[plane] + shift[plane]); grid.x = planes * chunks_per_plane
__global__ __launch_bounds__(256) void scale_shift_act_kernel(float *__restrict__ x, const float *__restrict__ scale,
                                                              const float *__restrict__ shift, int HW4, int chunks, int relu) {
  const unsigned plane = blockIdx.x / chunks;
  const int chunk = blockIdx.x - plane * chunks;
  const float a = scale[plane], b = shift[plane];
  float4 *xp = reinterpret_cast<float4 *>(x) + (size_t)plane * HW4;
  for (int i = chunk * 256 + threadIdx.x; i < HW4; i += chunks * 256) {
    float4 v = xp[i];
    v.x = v.x * a + b; v.y = v.y * a + b; v.z = v.z * a + b; v.w = v.w * a + b;
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    xp[i] = v;
  }
}

__global__ __launch_bounds__(256) void scale_shift_act_scalar_kernel(float *__restrict__ x, const float *__restrict__ scale,
                                                                     const float *__restrict__ shift, long long HW, long long total,
                                                                     int relu) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long plane = i / HW;
    const float v = x[i] * scale[plane] + shift[plane];
    x[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

}  // namespace

DVIS_EXPORT int dvis_group_norm_affine(const float *x, const float *gamma, const float *beta, float *scale, float *shift,
                                       int64_t N, int C, int G, int64_t HW, float eps, void *stream) {
  DVIS_REQUIRE(N >= 0 && C > 0 && G > 0 && HW > 0 && C % G == 0, "group_norm_affine: bad sizes (C=%d G=%d)", C, G);
  if (N == 0) return DVIS_OK;
  DVIS_REQUIRE(x && scale && shift, "group_norm_affine: null pointer");
  DVIS_REQUIRE(C / G <= 1024, "group_norm_affine: at most 1024 channels per group");
  const int vec = ((int64_t)(C / G) * HW) % 4 == 0 && ((uintptr_t)x & 15) == 0;
  DVIS_REQUIRE(N * G < (1ll << 31), "group_norm_affine: too many groups");
  hipLaunchKernelGGL(group_norm_affine_kernel, dim3((unsigned)(N * G)), dim3(1024), 0, (hipStream_t)stream, x, gamma, beta,
                     scale, shift, C, G, (long long)HW, eps, vec);
  return dvis_check_launch("group_norm_affine_kernel");
}

DVIS_EXPORT int dvis_scale_shift_act(float *x, const float *scale, const float *shift, int64_t planes, int64_t HW, int relu,
                                     void *stream) {
  DVIS_REQUIRE(planes >= 0 && HW > 0, "scale_shift_act: bad sizes");
  if (planes == 0) return DVIS_OK;
  DVIS_REQUIRE(x && scale && shift, "scale_shift_act: null pointer");
  if (HW % 4 != 0 || ((uintptr_t)x & 15) != 0) {
    const long long total = (long long)planes * HW;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(scale_shift_act_scalar_kernel, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, x, scale, shift, (long long)HW, total, relu);
    return dvis_check_launch("scale_shift_act_scalar_kernel");
  }
  const int HW4 = (int)(HW / 4);
  int chunks = (HW4 + 1023) / 1024;
  if (chunks < 1) chunks = 1;
  DVIS_REQUIRE(planes * chunks < (1ll << 31), "scale_shift_act: too many planes");
  hipLaunchKernelGGL(scale_shift_act_kernel, dim3((unsigned)(planes * chunks)), dim3(256), 0, (hipStream_t)stream, x, scale,
                     shift, HW4, chunks, relu);
  return dvis_check_launch("scale_shift_act_kernel");
}
