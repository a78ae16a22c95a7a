// Fast bilateral solver (Barron & Poole) for SelfMask mask refinement — utils/bilateral_solver.py:40-195 — in
// float64 on gfx950, plus the de-normalise-to-uint8 step in front of it (utils/utils.py:261-273).
// Compiled with -ffp-contract=off: bin edges are hit exactly (gray 16 -> Y = 15.999999999999998 -> bin 0) and the
// grid quantities (bistochastisation) are bit-identical to NumPy/SciPy.
//
// MI355X design.  The reference builds the grid with a float hash + np.unique (a sort of N keys) + five
// searchsorted passes + six CSR matrices.  Here the 5-D lattice is small and dense-indexable
// (Nx*Ny*Nl*Nu*Nv cells, 22.5 M at 512x683 => a 2.8 MB bitmap that lives in L2), and the dense cell order IS the
// hash order (hash = x + 255 y + 255^2 l + ... sorts by (v,u,l,y,x)).  So:
//   vertex id of a cell   = exclusive popcount-prefix of the occupancy bitmap   (== np.unique's sorted rank)
//   neighbour of a vertex = bit test at cell +/- stride_d + the same prefix      (== searchsorted + equality test)
// No sort, no hash table, no CSR.  Splat is deterministic and reproduces SciPy's CSR row sums bit for bit: pixel counts and
// (for uint8 targets) target sums are INTEGER atomics; splat(w) is `confidence` added count times; splat(t*w) is the same
// repeated add for binary targets and, for any other target, a per-vertex scan of the vertex's spatial cell in ascending
// pixel order (a vertex's pixels all lie in one sigma_spatial x sigma_spatial block).  Blur is a 10-entry gather summed in
// SciPy's CSR column order.  The PCG loop keeps every scalar (rho, alpha, ||r||, the convergence flag) on the device: 3
// launches per iteration, deterministic two-level reductions over the blocks that actually hold vertices (the block count
// is derived in-kernel from the vertex count), no host round trip until the result is sliced.
// Batching: every kernel takes blockIdx.y = image; the B images of a call (same H x W) have their own workspace slices, so
// the ~110 launches of a solve — launch/latency-bound for one image — are shared by the whole batch.
#include "common.h"
#include <stdlib.h>

typedef unsigned long long u64;

struct BgDims { int Nx, Ny, Nl, Nu, Nv; long cells; };

__device__ __forceinline__ double blur_gather(const double* __restrict__ x, const int* __restrict__ nb, double xv) {
  double out = 10.0 * xv;                                     // 2 * dim * x   (bilateral_solver.py:97)
  // selects, not branches: the ten gathers are requested together (absent neighbours read x[0] and are deselected) — same values, same
  // order of additions; these kernels are ~1 us of dependent latency each
  int j[10];
  double g[10];
#pragma unroll
  for (int e = 0; e < 10; ++e) j[e] = nb[e];
#pragma unroll
  for (int e = 0; e < 10; ++e) g[e] = x[j[e] >= 0 ? j[e] : 0];
#pragma unroll
  for (int d = 0; d < 5; ++d) {
    double t = j[2 * d] >= 0 ? g[2 * d] : 0.0;
    t = j[2 * d + 1] >= 0 ? t + g[2 * d + 1] : t;
    out = out + t;
  }
  return out;
}
// ---- de-normalise: fp32 x*std + mean, *255, clip, TRUNCATE (utils/utils.py:269-272).  x [3,H,W] -> rgb u8 [H,W,3]
__global__ __launch_bounds__(256) void denorm_kernel(const float* x, unsigned char* rgb, long HW, float m0, float m1, float m2,
                                                     float s0, float s1, float s2) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = __fadd_rn(__fmul_rn(x[c * HW + i], sd[c]), mean[c]);
    v = fminf(fmaxf(__fmul_rn(v, 255.0f), 0.0f), 255.0f);
    rgb[i * 3 + c] = (unsigned char)v;
  }
}

extern "C" int zh_denormalize_u8(const float* x, unsigned char* rgb, int H, int W, const float* mean3, const float* std3,
                                 hipStream_t stream) {
  ZH_CHECK_ARG(x && rgb && H > 0 && W > 0 && mean3 && std3, "zh_denormalize_u8: bad arguments");   // mean3/std3 are HOST pointers
  const long HW = (long)H * W;
  hipLaunchKernelGGL(denorm_kernel, dim3(zh_cdiv(HW, 256)), dim3(256), 0, stream, x, rgb, HW, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2]);
  ZH_CHECK_LAUNCH("zh_denormalize_u8");
  return ZH_OK;
}

// ---- batching: image = blockIdx.y; per-image strides of the caller's arrays and of the workspace
struct BgBatch { size_t ws; long N; long dbg; };     // workspace bytes per image, pixels per image, doubles per image in n_out / m_out
template <class T> __device__ __forceinline__ T* ws_img(T* p, const BgBatch& bt) { return (T*)((char*)p + (size_t)blockIdx.y * bt.ws); }

// ---- K1: 5-D coordinates (bilateral_solver.py:42-50) -> dense cell id, occupancy bitmap
__device__ __forceinline__ void yuv_bins(unsigned char R, unsigned char G, unsigned char B, double sl, double sc, int& l, int& u, int& v) {
  const double r = R, g = G, b = B;
  // np.tensordot -> OpenBLAS dgemm: a k-ordered FMA chain on every FMA-capable x86 (probed exhaustively over all 2^24
  // colours: plain mul/add order mis-bins 207 of them, the chain none).  Explicit fma(): the file is built with
  // -ffp-contract=off so nothing else is contracted.
  const double Y = fma(b, 0.114, fma(g, 0.587, r * 0.299));
  const double U = fma(b, 0.5, fma(g, -0.331264, r * -0.168736)) + 128.0;
  const double V = fma(b, -0.081312, fma(g, -0.418688, r * 0.5)) + 128.0;
  l = (int)(Y / sl); u = (int)(U / sc); v = (int)(V / sc);
}

// Runs of equal keys inside a wave (consecutive pixels of a row mostly share their lattice cell: 16 pixels per spatial
// cell, smooth colour): lane `l` is a run leader when its key differs from lane l-1's; returns the run length for leaders
// (0 for the other lanes) and the mask of the run's lanes.  One atomic per run instead of one per pixel — the per-pixel
// form spent ~0.9 ms per 8 images hammering the same 64-bit words / counters.
__device__ __forceinline__ int wave_run(int key, int lane, u64& run_mask) {
  const int prev = __shfl_up(key, 1, 64);
  const bool leader = lane == 0 || key != prev;
  const u64 lm = __ballot(leader);
  const u64 above = lane == 63 ? 0ull : (lm & ~((2ull << lane) - 1ull));
  const int next = above ? __ffsll((long long)above) - 1 : 64;
  const u64 upto = next == 64 ? ~0ull : ((1ull << next) - 1ull);
  run_mask = upto & ~((1ull << lane) - 1ull);
  return leader ? next - lane : 0;
}

__global__ __launch_bounds__(256) void bg_cells_kernel(const unsigned char* rgb, int H, int W, double ss, double sl, double sc,
                                                       BgDims dm, int* cell, u64* bitmap, int* coords_out, BgBatch bt) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool valid = p < (long)H * W;
  rgb += (size_t)blockIdx.y * bt.N * 3;
  int key = -1 - (int)(threadIdx.x & 63);                       // invalid lanes: unique keys, never a run with a neighbour
  long id = 0;
  if (valid) {
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    int l, u, v;
    yuv_bins(rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2], sl, sc, l, u, v);
    const int cx = (int)((double)x / ss), cy = (int)((double)y / ss);
    if (coords_out) { int* c = coords_out + 5 * ((size_t)blockIdx.y * bt.N + p); c[0] = cx; c[1] = cy; c[2] = l; c[3] = u; c[4] = v; }
    if (cell) {
      l = min(l, dm.Nl - 1); u = min(u, dm.Nu - 1); v = min(v, dm.Nv - 1);
      id = ((((long)v * dm.Nu + u) * dm.Nl + l) * dm.Ny + cy) * dm.Nx + cx;
      key = (int)id;
    }
  }
  if (!cell) return;                                            // block-uniform
  cell = ws_img(cell, bt); bitmap = ws_img(bitmap, bt);
  if (valid) cell[p] = (int)id;
  u64 rm;
  const int run = wave_run(key, threadIdx.x & 63, rm);
  if (valid && run > 0) {
    const u64 bit = 1ull << (id & 63);
    // idempotent: skip the atomic when the bit is already visible (a stale 0 only costs a redundant OR)
    if (!(__hip_atomic_load(&bitmap[id >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&bitmap[id >> 6], bit);
  }
}

// ---- K2: exclusive prefix of per-word popcounts (3 passes; words <= ~1M)
#define SCAN_PER_BLOCK 1024
__global__ __launch_bounds__(256) void bg_scan_block_sums(const u64* bitmap, long nwords, unsigned* blocksum, BgBatch bt) {
  bitmap = ws_img(bitmap, bt); blocksum = ws_img(blocksum, bt);
  const long base = (long)blockIdx.x * SCAN_PER_BLOCK;
  unsigned s = 0;
  for (int i = threadIdx.x; i < SCAN_PER_BLOCK; i += 256)
    if (base + i < nwords) s += __popcll(bitmap[base + i]);
  __shared__ unsigned red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) blocksum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void bg_scan_top(unsigned* blocksum, int nb, int* nvertices, BgBatch bt) {
  // one workgroup per image: exclusive scan of the nb (a few hundred) block sums, 256 at a time with a running carry
  blocksum = ws_img(blocksum, bt); nvertices = ws_img(nvertices, bt);
  __shared__ unsigned sh[256];
  __shared__ unsigned carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 256) {
    const int i = base + threadIdx.x;
    const unsigned mine = i < nb ? blocksum[i] : 0;
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const unsigned t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) blocksum[i] = carry + sh[threadIdx.x] - mine;
    __syncthreads();
    if (threadIdx.x == 0) carry += sh[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) *nvertices = (int)carry;
}
__global__ __launch_bounds__(256) void bg_scan_write(const u64* bitmap, long nwords, const unsigned* blocksum, unsigned* wprefix, BgBatch bt) {
  bitmap = ws_img(bitmap, bt); blocksum = ws_img(blocksum, bt); wprefix = ws_img(wprefix, bt);
  // each thread owns 4 consecutive words; block-level exclusive scan of the 256 thread sums
  const long base = (long)blockIdx.x * SCAN_PER_BLOCK + threadIdx.x * 4;
  unsigned c[4], s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { c[i] = base + i < nwords ? __popcll(bitmap[base + i]) : 0; s += c[i]; }
  __shared__ unsigned sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned run = blocksum[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
  for (int i = 0; i < 4; ++i) { if (base + i < nwords) wprefix[base + i] = run; run += c[i]; }
}

// ---- K3: pixel -> vertex; INTEGER splat of ones and (uint8 targets) of the target (S.dot, bilateral_solver.py:87-88)
__global__ __launch_bounds__(256) void bg_assign_kernel(const int* cell, const u64* bitmap, const unsigned* wprefix, long N,
                                                        const unsigned char* t_u8, int* pix2v, unsigned* cnt_i, unsigned* tsum_i,
                                                        unsigned* nonbinary, BgBatch bt) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  const bool valid = p < N;
  const int lane = threadIdx.x & 63;
  cell = ws_img(cell, bt); bitmap = ws_img(bitmap, bt); wprefix = ws_img(wprefix, bt); pix2v = ws_img(pix2v, bt);
  cnt_i = ws_img(cnt_i, bt); tsum_i = ws_img(tsum_i, bt); nonbinary = ws_img(nonbinary, bt);
  int v = -1 - lane;
  unsigned t = 0;
  if (valid) {
    const int id = cell[p];
    const u64 w = bitmap[id >> 6];
    v = (int)(wprefix[id >> 6] + __popcll(w & ((1ull << (id & 63)) - 1ull)));
    pix2v[p] = v;
    if (t_u8) t = t_u8[(size_t)blockIdx.y * bt.N + p];
  }
  // one integer atomic per run of equal vertices (integer sums: the grouping cannot change the result)
  u64 rm;
  const int run = wave_run(v, lane, rm);
  const u64 ones = __ballot(t != 0), big = __ballot(t > 1);
  if (valid && run > 0) atomicAdd(&cnt_i[v], (unsigned)run);
  if (t_u8) {
    if (big == 0) {                                              // wave-uniform: binary targets in this wave
      const int k = __popcll(ones & rm);
      if (valid && run > 0 && k) atomicAdd(&tsum_i[v], (unsigned)k);
    } else {
      if (valid && t) atomicAdd(&tsum_i[v], t);
      if (lane == 0) atomicOr(nonbinary, 1u);
    }
  }
}

// Vertex kernels are launched with a FIXED grid (VGRID blocks per image) and walk the 256-vertex blocks that actually exist
// (ceil(nv / 256), known only on the device): the worst case (one vertex per pixel) would need H*W/256 blocks per image and
// a typical image fills ~6 % of them — the idle blocks were most of the dispatch cost of every launch.
#define VGRID 192
#define FOR_VERTEX_BLOCKS(vb, nvp) for (int vb = blockIdx.x, vb##_n = (*(nvp) + 255) >> 8; vb < vb##_n; vb += gridDim.x)
#define FOR_VERTEX_BLOCKS_G(vb, nvp, g, G) for (int vb = (g), vb##_n = (*(nvp) + 255) >> 8; vb < vb##_n; vb += (G))

// ---- K4/K5: vertex -> cell, neighbour table (get_valid_idx, bilateral_solver.py:29-37,69-81)
__global__ __launch_bounds__(256) void bg_vertices_kernel(const u64* bitmap, const unsigned* wprefix, long nwords, int* vcell, BgBatch bt) {
  const long w = (long)blockIdx.x * 256 + threadIdx.x;
  if (w >= nwords) return;
  bitmap = ws_img(bitmap, bt); wprefix = ws_img(wprefix, bt); vcell = ws_img(vcell, bt);
  u64 bits = bitmap[w];
  unsigned v = wprefix[w];
  while (bits) {
    const int b = __ffsll((long long)bits) - 1;
    vcell[v++] = (int)(w * 64 + b);
    bits &= bits - 1;
  }
}
__global__ __launch_bounds__(256) void bg_neighbors_kernel(const int* vcell, const u64* bitmap, const unsigned* wprefix, const int* nv,
                                                           BgDims dm, int* nbr, BgBatch bt) {
  vcell = ws_img(vcell, bt); bitmap = ws_img(bitmap, bt); wprefix = ws_img(wprefix, bt); nv = ws_img(nv, bt); nbr = ws_img(nbr, bt);
  FOR_VERTEX_BLOCKS(vb, nv) {
  const int v = vb * 256 + threadIdx.x;
  if (v >= *nv) continue;
  const long id = vcell[v];
  long r = id;
  int c[5];
  const int dims[5] = {dm.Nx, dm.Ny, dm.Nl, dm.Nu, dm.Nv};
  long stride[5];
  long st = 1;
#pragma unroll
  for (int d = 0; d < 5; ++d) { c[d] = (int)(r % dims[d]); r /= dims[d]; stride[d] = st; st *= dims[d]; }
#pragma unroll
  for (int d = 0; d < 5; ++d)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int cc = c[d] + (s ? 1 : -1);
      int out = -1;
      if (cc >= 0 && cc < dims[d]) {
        const long j = id + (s ? stride[d] : -stride[d]);
        const u64 w = bitmap[j >> 6];
        if ((w >> (j & 63)) & 1ull) out = (int)(wprefix[j >> 6] + __popcll(w & ((1ull << (j & 63)) - 1ull)));
      }
      nbr[v * 10 + 2 * d + s] = out;
    }
  }
}

// ---- splat finalisation, one thread per vertex: the float64 sums SciPy's CSR matvec forms, in its order.
//   cnt    = S.1            (exact integer)
//   wsplat = S.(conf * 1)   = conf added cnt times
//   bsplat = S.(t * conf)   = binary targets: conf added (number of ones) times (x + 0.0 == x);
//                             otherwise the vertex's spatial cell is scanned in ascending pixel order
__global__ __launch_bounds__(256) void bg_splat_final_kernel(const unsigned* cnt_i, const unsigned* tsum_i, const unsigned* nonbinary,
                                                             const int* pix2v, const int* vcell, const int* nv, const unsigned char* t_u8,
                                                             const double* t_f64, int H, int W, double ss, BgDims dm, double conf,
                                                             double* cnt, double* wsplat, double* bsplat, BgBatch bt) {
  cnt_i = ws_img(cnt_i, bt); tsum_i = ws_img(tsum_i, bt); nonbinary = ws_img(nonbinary, bt); pix2v = ws_img(pix2v, bt);
  vcell = ws_img(vcell, bt); nv = ws_img(nv, bt); cnt = ws_img(cnt, bt); wsplat = ws_img(wsplat, bt); bsplat = ws_img(bsplat, bt);
  // conf added j times, IN ORDER (the very chain the per-vertex loops below walked: a vertex holds up to a spatial cell's pixels, a few
  // hundred, and a wave waited for its longest chain twice — 17.5 us of a 0.27-ms solve).  One thread walks the chain once per block into
  // LDS; every vertex looks its two sums up.  Counts beyond the table fall back to the loop.
  constexpr int SPLAT_TAB = 320;      // a spatial cell holds up to (sigma_spatial + 2)^2 = 324 pixels at the reference's sigma_spatial = 16: counts 321 .. 324 take the loop below
  __shared__ double conf_times[SPLAT_TAB + 1];
  if (threadIdx.x == 0) {
    double a = 0.0;
    conf_times[0] = 0.0;
    for (int j = 1; j <= SPLAT_TAB; ++j) { a = a + conf; conf_times[j] = a; }
  }
  __syncthreads();
  auto conf_sum = [&](unsigned k) {
    if (k <= (unsigned)SPLAT_TAB) return conf_times[k];
    double a = conf_times[SPLAT_TAB];
    for (unsigned i = SPLAT_TAB; i < k; ++i) a = a + conf;
    return a;
  };
  FOR_VERTEX_BLOCKS(vb, nv) {
  const int v = vb * 256 + threadIdx.x;
  if (v >= *nv) continue;
  const unsigned k = cnt_i[v];
  cnt[v] = (double)k;
  wsplat[v] = conf_sum(k);
  double sb = 0.0;
  if (t_u8 && *nonbinary == 0) {
    sb = conf_sum(tsum_i[v]);
  } else {
    const long id = vcell[v];
    const int cx = (int)(id % dm.Nx), cy = (int)((id / dm.Nx) % dm.Ny);
    const int x0 = max(0, (int)((double)cx * ss) - 1), x1 = min(W - 1, (int)((double)(cx + 1) * ss) + 1);
    const int y0 = max(0, (int)((double)cy * ss) - 1), y1 = min(H - 1, (int)((double)(cy + 1) * ss) + 1);
    const size_t ib = (size_t)blockIdx.y * bt.N;
    for (int y = y0; y <= y1; ++y)
      for (int x = x0; x <= x1; ++x) {
        const long p = (long)y * W + x;
        if (pix2v[p] == v) sb = sb + (t_u8 ? (double)t_u8[ib + p] : t_f64[ib + p]) * conf;
      }
  }
  bsplat[v] = sb;
  }
}

// ---- bistochastisation (bilateral_solver.py:107-118)
__global__ __launch_bounds__(256) void bg_bisto_step(const double* n_in, const double* m0, const int* nbr, const int* nv, double* n_out, BgBatch bt) {
  n_in = ws_img(n_in, bt); m0 = ws_img(m0, bt); nbr = ws_img(nbr, bt); nv = ws_img(nv, bt); n_out = ws_img(n_out, bt);
  FOR_VERTEX_BLOCKS(vb, nv) {
    const int v = vb * 256 + threadIdx.x;
    if (v >= *nv) continue;
    const double nvv = n_in[v];
    n_out[v] = sqrt(nvv * m0[v] / blur_gather(n_in, nbr + v * 10, nvv));
  }
}
// the first step, from n = 1 (bilateral_solver.py:110): 1 * m0 / blur(ones) — the same operations on the constant, without the fill
// launch that used to write the ones (blur(ones) = 10 + the number of neighbours that exist: small integers, exact in any order)
__global__ __launch_bounds__(256) void bg_bisto_first(const double* m0, const int* nbr, const int* nv, double* n_out, BgBatch bt) {
  m0 = ws_img(m0, bt); nbr = ws_img(nbr, bt); nv = ws_img(nv, bt); n_out = ws_img(n_out, bt);
  FOR_VERTEX_BLOCKS(vb, nv) {
    const int v = vb * 256 + threadIdx.x;
    if (v >= *nv) continue;
    const int* nb = nbr + v * 10;
    double out = 10.0 * 1.0;
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      double t = nb[2 * d] >= 0 ? 1.0 : 0.0;
      if (nb[2 * d + 1] >= 0) t = t + 1.0;
      out = out + t;
    }
    n_out[v] = sqrt(1.0 * m0[v] / out);
  }
}
// ---- PCG (scipy.sparse.linalg.cg semantics, Jacobi preconditioner; bilateral_solver.py:133-147), single-reduction form
//
// Round 5.  A solve at 512 x 683 is ~20 k vertices: every launch is ~1 us of work behind ~5 us of launch latency, so the solve costs
// what its LAUNCH COUNT costs.  The textbook loop SciPy runs (z = M^-1 r; rho = r.z; p = z + beta p; q = A p; alpha = rho / p.q;
// x += alpha p; r -= alpha q) has two grid-wide reductions per iteration, i.e. two launches (rounds 3 - 4: 2 + 2 x 25 launches).
// The Chronopoulos - Gear rearrangement has ONE: with w = A z, gamma = r.z and delta = w.z taken in the same sweep,
//     beta = gamma / gamma_prev,  alpha = gamma / (delta - beta gamma / alpha_prev),
//     p = z + beta p,  s = w + beta s  (= A p),  x += alpha p,  r -= alpha s
// — the same iterates in exact arithmetic (p.q = delta - beta gamma / alpha_prev), different rounding (measured against the reference's
// goldens: tests/test_bilateral_gpu.py holds 1e-9 on the soft output, equality on the > 0.5 mask and on the iteration counts).
// ONE launch per iteration: kernel K(it) first reduces the partials K(it - 1) left (gamma, delta, ||r||^2 — every block re-derives the
// scalars from the same partials in the same order: deterministic, no flag race), tests convergence as SciPy does at the top of
// iteration `it`, then for its vertex AND, on the fly, for the vertex's (up to) ten neighbours forms s, r, z of the NEW iterate (the
// same uncontracted expressions: every copy of a value is the same double), applies A to the new z, and leaves the next partials.
// State {r, w, s, p} per vertex is one 32-byte record, ping-ponged between two buffers by iteration parity (a thread reads its
// neighbours' old records while others write new ones); {1 / diag, n} is a 16-byte record.  K(-1) starts the chain (w = A z of the
// initial residual; no update), K(maxiter - 1) only finishes x: maxiter + 1 launches, 26 for the reference's 25 iterations.
struct __attribute__((aligned(32))) CgDyn { double r, w, s, p; };
struct __attribute__((aligned(16))) CgCst { double minv, n; };
struct CgPtrs {
  const double *n, *m, *wsplat, *b; const int* nbr; const int* nv;
  double* mw;                            // m again, writable: the init kernel computes it
  CgCst* cst; CgDyn *dyn0, *dyn1; double* x;
  double* part;                          // per-block partial sums: 2 sets (iteration parity) x {gamma, delta, rr}; set 0 first holds b.b
  int nblocks;
  double* sc;                            // scalars: [0],[1] gamma ping-pong, [2] atol, [3] converged flag, [4] iterations, [5],[6] alpha ping-pong
  double lam, a_diag_min, rtol;
};
__device__ __forceinline__ CgPtrs cg_img(CgPtrs c, const BgBatch& bt) {
  c.n = ws_img(c.n, bt); c.m = ws_img(c.m, bt); c.wsplat = ws_img(c.wsplat, bt); c.b = ws_img(c.b, bt); c.nbr = ws_img(c.nbr, bt);
  c.nv = ws_img(c.nv, bt); c.mw = ws_img(c.mw, bt); c.cst = ws_img(c.cst, bt); c.dyn0 = ws_img(c.dyn0, bt); c.dyn1 = ws_img(c.dyn1, bt); c.x = ws_img(c.x, bt);
  c.part = ws_img(c.part, bt); c.sc = ws_img(c.sc, bt);
  return c;
}
// blocks that hold vertices: the launch grid is sized for the worst case (one vertex per pixel); everything past this exits
__device__ __forceinline__ int cg_active_blocks(const CgPtrs& c) { return (*c.nv + 255) >> 8; }

__device__ __forceinline__ double block_sum(double s, double* red) {   // deterministic: fixed tree, fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
// three sums at once: one LDS exchange and two barriers instead of three of each (a step kernel is ~1 us of work: every barrier shows)
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; red[w] = a; red[4 + w] = b; red[8 + w] = c; }
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  b = (red[4] + red[5]) + (red[6] + red[7]);
  c = (red[8] + red[9]) + (red[10] + red[11]);
}
__device__ __forceinline__ double sum_partials(const double* part, int nb, double* red) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  return block_sum(s, red);
}

// x0 = splat(xw) / splat(w) (the flat initialisation, bilateral_solver.py:141 — formed for the vertex and, on the fly, for its
// neighbours: the same division, the same double), r0 = b - A x0, the Jacobi preconditioner, per-block partials of b.b (for atol)
__global__ __launch_bounds__(256) void cg_init_kernel(CgPtrs c0, BgBatch bt) {
  __shared__ double red[4];
  const CgPtrs c = cg_img(c0, bt);
  FOR_VERTEX_BLOCKS(vb, c.nv) {
    const int v = vb * 256 + threadIdx.x;
    double bb = 0.0;
    if (v < *c.nv) {
      // m = n .* blur(n) (bilateral_solver.py:114; only the vertex's own m is needed below: it was a launch of its own)
      const double mv = c.n[v] * blur_gather(c.n, c.nbr + v * 10, c.n[v]);
      c.mw[v] = mv;
      const double diag = c.lam * (mv - c.n[v] * 10.0 * c.n[v]) + c.wsplat[v];
      const double mi = 1.0 / fmax(diag, c.a_diag_min);
      c.cst[v] = CgCst{mi, c.n[v]};
      auto xval = [&](int j) { return c.b[j] / c.wsplat[j]; };
      const double xv = xval(v);
      c.x[v] = xv;
      const int* nb10 = c.nbr + v * 10;
      const double ny = c.n[v] * xv;
      double bl = 10.0 * ny;
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const int lo = nb10[2 * d], hi = nb10[2 * d + 1];
        double t = lo >= 0 ? c.n[lo] * xval(lo) : 0.0;
        if (hi >= 0) t = t + c.n[hi] * xval(hi);
        bl = bl + t;
      }
      const double r = c.b[v] - (c.lam * (mv * xv - c.n[v] * bl) + c.wsplat[v] * xv);
      c.dyn0[v] = CgDyn{r, 0.0, 0.0, 0.0};
      bb = c.b[v] * c.b[v];
    }
    const double s3 = block_sum(bb, red);
    if (threadIdx.x == 0) c.part[vb] = s3;
  }
}
// K(it), it = -1 .. maxiter - 1 (see the header of this section)
// Everything a vertex's update reads, requested BEFORE the kernel's scalars are known (none of it depends on alpha / beta): a step is a
// chain of dependent memory round trips (partials -> scalars | indices -> records -> sums), and with the vertex's operands in flight
// under the reduction of the partials the chain is two trips shorter (7.8 -> see profiles/NOTES.md round 5, us per launch at 20 k vertices).
struct CgVtxIn { bool ok; CgCst cv; CgDyn dv; double mv, wv, xv; int nbi[10]; CgCst cj[10]; double rj[10], wj[10], sj[10]; };
__global__ __launch_bounds__(256) void cg_step_kernel(CgPtrs c0, int it, int last, BgBatch bt) {
  __shared__ double red[12];
  const CgPtrs c = cg_img(c0, bt);
  const int nv = *c.nv, nb = (nv + 255) >> 8;
  if ((int)blockIdx.x >= nb) return;
  const bool plain = it <= 0;                                               // no previous direction: s = w, p = z
  const CgDyn* __restrict__ dold = ((it + 1) & 1) ? c.dyn1 : c.dyn0;
  CgDyn* __restrict__ dnew = ((it + 1) & 1) ? c.dyn0 : c.dyn1;
  const CgCst* __restrict__ cst = c.cst;
  auto load_vtx = [&](int v, CgVtxIn& in) {
    in.ok = v < nv;
    const int vv = in.ok ? v : 0;                                           // surplus threads of the last block read vertex 0 and store nothing
    in.cv = cst[vv]; in.dv = dold[vv]; in.mv = c.m[vv]; in.wv = c.wsplat[vv]; in.xv = c.x[vv];
    const int* nb10 = c.nbr + (size_t)vv * 10;
#pragma unroll
    for (int e = 0; e < 10; ++e) in.nbi[e] = nb10[e];
    if (!last) {
#pragma unroll
      for (int e = 0; e < 10; ++e) {                                        // selects, not branches: absent neighbours read the vertex's own
        const int j = in.nbi[e] >= 0 ? in.nbi[e] : vv;                      // record and are deselected in the sum — 30 loads in flight together
        in.cj[e] = cst[j];
        const CgDyn dj = dold[j];
        in.rj[e] = dj.r; in.wj[e] = dj.w; in.sj[e] = dj.s;
      }
    }
  };
  int vb = blockIdx.x;
  CgVtxIn in;
  load_vtx(vb * 256 + threadIdx.x, in);
  // ---- the iteration's scalars from the partials K(it - 1) left (it = -1: the init kernel's b.b)
  double alpha = 0.0, beta = 0.0;
  const double* pin = c.part + (size_t)((it + 1) & 1) * 3 * c.nblocks;
  double* pout = c.part + (size_t)(it & 1) * 3 * c.nblocks;
  const double flag = c.sc[3], atol = c.sc[2], bzero = c.sc[7];               // all written by EARLIER kernels only
  const double gp = c.sc[it >= 1 ? ((it - 1) & 1) : 0], ap = c.sc[it >= 1 ? 5 + ((it - 1) & 1) : 5];
  if (it < 0) {
    const double bb = sum_partials(pin, nb, red);                           // atol = rtol * ||b||
    if (blockIdx.x == 0 && threadIdx.x == 0) { c.sc[2] = c.rtol * sqrt(bb); c.sc[7] = bb == 0.0 ? 1.0 : 0.0; }
  } else {
    if (flag != 0.0) return;                                                // converged earlier
    double gamma = 0.0, delta = 0.0, rr = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) { gamma += pin[i]; delta += pin[c.nblocks + i]; rr += pin[2 * c.nblocks + i]; }
    block_sum3(gamma, delta, rr, red);
    // SciPy's test at the top of iteration `it`; and its `if bnrm2 == 0: return b` (||b|| = 0: an EMPTY target — the
    // pseudo-labeller found nothing — solves to zeros without iterating; x0 = b / w is already 0 there, where the loop would form 0 / 0).
    // The flag is ||b|| == 0 itself, recorded by K(-1): atol = rtol ||b|| is ALSO zero for cg_tol = 0 (or an underflowing product),
    // where SciPy iterates to maxiter (round-5 advisor)
    if (sqrt(rr) < atol || bzero != 0.0) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { c.sc[3] = 1.0; c.sc[4] = (double)it; }
      return;
    }
    if (it == 0) { beta = 0.0; alpha = gamma / delta; }
    else {
      beta = gamma / gp;
      alpha = gamma / (delta - beta * gamma / ap);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { c.sc[it & 1] = gamma; c.sc[5 + (it & 1)] = alpha; }
  }
  for (;;) {
    const int v = vb * 256 + threadIdx.x;
    double g = 0.0, d = 0.0, rr = 0.0;
    if (in.ok) {
      const double z = in.cv.minv * in.dv.r;
      const double sn = plain ? in.dv.w : in.dv.w + beta * in.dv.s;
      const double pn = plain ? z : z + beta * in.dv.p;
      const double rn = in.dv.r - alpha * sn;
      const double zn = in.cv.minv * rn;
      if (it >= 0) c.x[v] = in.xv + alpha * pn;
      if (!last) {
        // w = A z_new = lam * (m z - n * blur(n .* z)) + wsplat z; a neighbour's n z_new from its OLD record, by the expressions of
        // the vertex's own update above, term for term (every copy of a value is the same double)
        const double ny = in.cv.n * zn;
        double bl = 10.0 * ny;
        double nzv[10];
#pragma unroll
        for (int e = 0; e < 10; ++e) {
          const double sj = plain ? in.wj[e] : in.wj[e] + beta * in.sj[e];
          const double rj = in.rj[e] - alpha * sj;
          nzv[e] = in.cj[e].n * (in.cj[e].minv * rj);
        }
#pragma unroll
        for (int e = 0; e < 5; ++e) {                                       // same values, same order of additions as blur_gather
          double t = in.nbi[2 * e] >= 0 ? nzv[2 * e] : 0.0;
          t = in.nbi[2 * e + 1] >= 0 ? t + nzv[2 * e + 1] : t;
          bl = bl + t;
        }
        const double wn = c.lam * (in.mv * zn - in.cv.n * bl) + in.wv * zn;
        dnew[v] = CgDyn{rn, wn, sn, pn};
        g = rn * zn; d = wn * zn; rr = rn * rn;
      }
    }
    if (!last) {
      block_sum3(g, d, rr, red);
      if (threadIdx.x == 0) { pout[vb] = g; pout[c.nblocks + vb] = d; pout[2 * c.nblocks + vb] = rr; }
    }
    vb += gridDim.x;                                                        // more than VGRID blocks of vertices: noise images only
    if (vb >= nb) break;
    load_vtx(vb * 256 + threadIdx.x, in);
  }
}

// (Round 3 measured two persistent single-launch forms of the two-reduction loop — all iterations in one kernel with image-local barriers
// between the phases — and kept neither; both are in the git history.  With portable agent-scope release / acquire fences, as
// cooperative-groups grid sync does, a barrier costs ~26 us on this 8-XCD part (L2 write-back + invalidate): 2.15 ms per
// 512x683 solve against 0.49 ms for the 75 separate launches.  With one image per XCD (blocks b, b+8, ... share an XCD; placement
// verified through HW_REG_XCC_ID) and barriers that stay inside that XCD's L2 (s_waitcnt vmcnt(0) + an L2-executed atomic + poll +
// buffer_inv sc0) a barrier still costs ~5 us with 32 blocks polling one line — the price of a launch: 0.538 vs 0.477 ms at batch 1,
// 0.711 vs 0.740 ms at batch 8.  A phase is ~1 us of work on 20 k vertices; the solve is bound by the synchronisation latency
// itself, whichever mechanism provides it — so round 5 removed synchronisation POINTS instead: the single-reduction loop above.)

// debug copies of the bistochastisation vectors (tests only: launched when n_out / m_out are given)
__global__ void cg_debug_copy_kernel(const int* nv, const double* n_src, const double* m_src, double* n_out, double* m_out, BgBatch bt) {
  nv = ws_img(nv, bt); n_src = ws_img(n_src, bt); m_src = ws_img(m_src, bt);
  for (int v = threadIdx.x; v < *nv; v += blockDim.x) {
    if (n_out) n_out[(size_t)blockIdx.y * bt.dbg + v] = n_src[v];
    if (m_out) m_out[(size_t)blockIdx.y * bt.dbg + v] = m_src[v];
  }
}

// S^T y (bilateral_solver.py:148) + the solve's statistics {vertices, iterations} (one thread of the image's first block: it was a
// launch of its own)
__global__ __launch_bounds__(256) void bg_slice_kernel(const double* y, const int* pix2v, long N, double* out, const int* nv, const double* sc,
                                                       int maxiter, int* stats, BgBatch bt) {
  y = ws_img(y, bt); pix2v = ws_img(pix2v, bt);
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) {
    nv = ws_img(nv, bt); sc = ws_img(sc, bt);
    stats[2 * blockIdx.y] = *nv;
    stats[2 * blockIdx.y + 1] = sc[3] == 0.0 ? maxiter : (int)sc[4];
  }
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p < N) out[(size_t)blockIdx.y * bt.N + p] = y[pix2v[p]];
}

// ---- host orchestration -----------------------------------------------------------------------------------------
static BgDims bg_dims(int H, int W, double ss, double sl, double sc) {
  BgDims d;
  d.Nx = (int)((double)(W - 1) / ss) + 1; d.Ny = (int)((double)(H - 1) / ss) + 1;
  d.Nl = (int)(255.0 / sl) + 1; d.Nu = (int)(255.5 / sc) + 1; d.Nv = d.Nu;
  d.cells = (long)d.Nx * d.Ny * d.Nl * d.Nu * d.Nv;
  return d;
}
static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct BgLayout {
  size_t bitmap, wprefix, blocksum, cell, pix2v, vcell, nbr, ints, dbl, part, sc, nv, total;
  long nwords; int nscan; long Vmax; int nblocks;
};
static BgLayout bg_layout(int H, int W, const BgDims& d) {
  BgLayout L;
  const long N = (long)H * W;
  L.nwords = (d.cells + 63) / 64;
  L.nscan = zh_cdiv(L.nwords, SCAN_PER_BLOCK);
  L.Vmax = N < d.cells ? N : d.cells;
  L.nblocks = zh_cdiv(L.Vmax, 256);
  size_t o = 0;
  L.bitmap = o; o += al((size_t)L.nwords * 8);
  L.wprefix = o; o += al((size_t)L.nwords * 4);
  L.blocksum = o; o += al((size_t)L.nscan * 4);
  L.cell = o; o += al((size_t)N * 4);
  L.pix2v = o; o += al((size_t)N * 4);
  L.vcell = o; o += al((size_t)L.Vmax * 4);
  L.nbr = o; o += al((size_t)L.Vmax * 40);
  L.ints = o; o += al((size_t)L.Vmax * 4) * 2 + 256;  // cnt_i, tsum_i, nonbinary flag
  L.dbl = o; o += al((size_t)L.Vmax * 8) * 17;      // cnt, wsplat, bsplat, nA, nB, m, {minv, n} records (2), {r, w, s, p} records x 2 (8), x
  L.part = o; o += al((size_t)L.nblocks * 8 * 6);   // two sets (iteration parity) of {gamma, delta, rr} per 256-vertex block
  L.sc = o; o += 256;
  L.nv = o; o += 256;
  L.total = al(o);
  return L;
}

extern "C" size_t zh_bilateral_workspace_size(int H, int W, double sigma_spatial, double sigma_luma, double sigma_chroma) {
  if (H <= 0 || W <= 0 || sigma_spatial <= 0 || sigma_luma <= 0 || sigma_chroma <= 0) return 0;
  const BgDims d = bg_dims(H, W, sigma_spatial, sigma_luma, sigma_chroma);
  return bg_layout(H, W, d).total;
}

// debug / parity entry: per-pixel integer coordinates (x, y, luma, u, v) exactly as bilateral_solver.py:42-50
extern "C" int zh_bgrid_coords(const unsigned char* rgb, int H, int W, double sigma_spatial, double sigma_luma, double sigma_chroma,
                               int* coords, hipStream_t stream) {
  ZH_CHECK_ARG(rgb && coords && H > 0 && W > 0, "zh_bgrid_coords: bad arguments");
  BgDims d = bg_dims(H, W, sigma_spatial, sigma_luma, sigma_chroma);
  const BgBatch bt{0, (long)H * W, 0};
  hipLaunchKernelGGL(bg_cells_kernel, dim3(zh_cdiv((long)H * W, 256)), dim3(256), 0, stream, rgb, H, W, sigma_spatial, sigma_luma,
                     sigma_chroma, d, (int*)nullptr, (u64*)nullptr, coords, bt);
  ZH_CHECK_LAUNCH("zh_bgrid_coords");
  return ZH_OK;
}

// Whole solver for a batch of B images of one size, one channel each.  rgb u8 [B,H,W,3]; target u8 [B,H,W] (target_u8) or f64
// [B,H,W] (target_f64), exactly one non-NULL; out_soft f64 [B,H,W].  stats (device, may be NULL): int32 [B,2] = {nvertices,
// cg iterations}.  Optional debug outputs (device, may be NULL): n_out / m_out f64 [B, H*W] (first nvertices entries of
// each row).  workspace: B * zh_bilateral_workspace_size(H, W, ...) bytes.
__global__ __launch_bounds__(256) void bg_zero_kernel(u64* a, long na, u64* b, long nb, u64* c, long nc, BgBatch bt) {
  a = ws_img(a, bt); b = ws_img(b, bt); c = ws_img(c, bt);
  const long stride = (long)gridDim.x * 256, i0 = (long)blockIdx.x * 256 + threadIdx.x;
  for (long i = i0; i < na; i += stride) a[i] = 0;
  for (long i = i0; i < nb; i += stride) b[i] = 0;
  for (long i = i0; i < nc; i += stride) c[i] = 0;
}

extern "C" int zh_bilateral_solve_batch(const unsigned char* rgb, const unsigned char* target_u8, const double* target_f64, int B, int H,
                                        int W, double sigma_spatial, double sigma_luma, double sigma_chroma, double confidence,
                                        double lam, double a_diag_min, double cg_tol, int cg_maxiter, double* out_soft, int* stats,
                                        double* n_out, double* m_out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  ZH_CHECK_ARG(rgb && out_soft && B > 0 && B < 65536 && H > 0 && W > 0, "zh_bilateral_solve: bad arguments");
  ZH_CHECK_ARG((target_u8 != nullptr) != (target_f64 != nullptr), "zh_bilateral_solve: pass exactly one of target_u8 / target_f64");
  ZH_CHECK_ARG(sigma_spatial > 0 && sigma_luma > 0 && sigma_chroma > 0 && cg_maxiter >= 0, "zh_bilateral_solve: bad parameters");
  const BgDims d = bg_dims(H, W, sigma_spatial, sigma_luma, sigma_chroma);
  ZH_CHECK_ARG(d.cells < (1L << 31), "zh_bilateral_solve: lattice too large (%ld cells)", d.cells);
  const BgLayout L = bg_layout(H, W, d);
  if (!workspace || workspace_bytes < L.total * (size_t)B) {
    zh_set_error("zh_bilateral_solve: workspace too small (%zu < %zu)", workspace_bytes, L.total * (size_t)B);
    return ZH_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  const long N = (long)H * W;
  const BgBatch bt{L.total, N, N};
  u64* bitmap = (u64*)(ws + L.bitmap);
  unsigned* wprefix = (unsigned*)(ws + L.wprefix);
  unsigned* blocksum = (unsigned*)(ws + L.blocksum);
  int* cell = (int*)(ws + L.cell);
  int* pix2v = (int*)(ws + L.pix2v);
  int* vcell = (int*)(ws + L.vcell);
  int* nbr = (int*)(ws + L.nbr);
  const size_t istride = al((size_t)L.Vmax * 4);
  unsigned* cnt_i = (unsigned*)(ws + L.ints);
  unsigned* tsum_i = (unsigned*)(ws + L.ints + istride);
  unsigned* nonbin = (unsigned*)(ws + L.ints + 2 * istride);
  const size_t dstride = al((size_t)L.Vmax * 8);
  double* D[17];
  for (int i = 0; i < 17; ++i) D[i] = (double*)(ws + L.dbl + dstride * i);
  double *cnt = D[0], *wsplat = D[1], *bsplat = D[2], *nA = D[3], *nB = D[4], *m = D[5];
  double* part = (double*)(ws + L.part);
  double* sc = (double*)(ws + L.sc);
  int* nv = (int*)(ws + L.nv);

  const dim3 blk(256), gN(zh_cdiv(N, 256), B), gV(L.nblocks < VGRID ? L.nblocks : VGRID, B), g1(1, B);
  // bitmaps, integer splats, scalars: ONE launch for the whole batch
  // (it was three hipMemsetAsync per image: 120 fill kernels per 5 batched calls, 7.5 % of the GPU time in the round-2 profile)
  hipLaunchKernelGGL(bg_zero_kernel, dim3(64, B), blk, 0, stream, (u64*)(ws + L.bitmap), L.nwords, (u64*)(ws + L.ints), (long)((2 * istride + 256) / 8),
                     (u64*)(ws + L.sc), 32L, bt);
  hipLaunchKernelGGL(bg_cells_kernel, gN, blk, 0, stream, rgb, H, W, sigma_spatial, sigma_luma, sigma_chroma, d, cell, bitmap, (int*)nullptr, bt);
  hipLaunchKernelGGL(bg_scan_block_sums, dim3(L.nscan, B), blk, 0, stream, bitmap, L.nwords, blocksum, bt);
  hipLaunchKernelGGL(bg_scan_top, g1, blk, 0, stream, blocksum, L.nscan, nv, bt);
  hipLaunchKernelGGL(bg_scan_write, dim3(L.nscan, B), blk, 0, stream, bitmap, L.nwords, blocksum, wprefix, bt);
  hipLaunchKernelGGL(bg_assign_kernel, gN, blk, 0, stream, cell, bitmap, wprefix, N, target_u8, pix2v, cnt_i, tsum_i, nonbin, bt);
  hipLaunchKernelGGL(bg_vertices_kernel, dim3(zh_cdiv(L.nwords, 256), B), blk, 0, stream, bitmap, wprefix, L.nwords, vcell, bt);
  hipLaunchKernelGGL(bg_neighbors_kernel, gV, blk, 0, stream, vcell, bitmap, wprefix, nv, d, nbr, bt);
  hipLaunchKernelGGL(bg_splat_final_kernel, gV, blk, 0, stream, cnt_i, tsum_i, nonbin, pix2v, vcell, nv, target_u8, target_f64, H, W,
                     sigma_spatial, d, confidence, cnt, wsplat, bsplat, bt);
  // bistochastise: n = 1; 10x n = sqrt(n*m0/blur(n)); m = n*blur(n)
  hipLaunchKernelGGL(bg_bisto_first, gV, blk, 0, stream, cnt, nbr, nv, nA, bt);
  double *ncur = nA, *nnext = nB;
  for (int i = 1; i < 10; ++i) {
    hipLaunchKernelGGL(bg_bisto_step, gV, blk, 0, stream, ncur, cnt, nbr, nv, nnext, bt);
    double* t = ncur; ncur = nnext; nnext = t;
  }
  // (m = n .* blur(n) is formed by the PCG's init kernel)
  // PCG
  CgPtrs c;
  c.n = ncur; c.m = m; c.mw = m; c.wsplat = wsplat; c.b = bsplat; c.nbr = nbr; c.nv = nv;
  c.cst = (CgCst*)D[6]; c.dyn0 = (CgDyn*)D[8]; c.dyn1 = (CgDyn*)D[12]; c.x = D[16];     // dstride is a multiple of 256 bytes: records stay aligned
  c.part = part; c.nblocks = L.nblocks; c.sc = sc;
  c.lam = lam; c.a_diag_min = a_diag_min; c.rtol = cg_tol;
  hipLaunchKernelGGL(cg_init_kernel, gV, blk, 0, stream, c, bt);
  for (int it = -1; it < cg_maxiter; ++it)
    hipLaunchKernelGGL(cg_step_kernel, gV, blk, 0, stream, c, it, (int)(it == cg_maxiter - 1), bt);
  hipLaunchKernelGGL(bg_slice_kernel, gN, blk, 0, stream, c.x, pix2v, N, out_soft, nv, sc, cg_maxiter, stats, bt);
  if (n_out || m_out) hipLaunchKernelGGL(cg_debug_copy_kernel, g1, blk, 0, stream, nv, ncur, m, n_out, m_out, bt);
  ZH_CHECK_LAUNCH("zh_bilateral_solve");
  return ZH_OK;
}

// One image (the reference's call, utils/bilateral_solver.py:152-195): the batch entry with B = 1.
extern "C" int zh_bilateral_solve(const unsigned char* rgb, const unsigned char* target_u8, const double* target_f64, int H, int W,
                                  double sigma_spatial, double sigma_luma, double sigma_chroma, double confidence, double lam,
                                  double a_diag_min, double cg_tol, int cg_maxiter, double* out_soft, int* stats,
                                  double* n_out, double* m_out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return zh_bilateral_solve_batch(rgb, target_u8, target_f64, 1, H, W, sigma_spatial, sigma_luma, sigma_chroma, confidence, lam, a_diag_min,
                                  cg_tol, cg_maxiter, out_soft, stats, n_out, m_out, workspace, workspace_bytes, stream);
}

// ---- output_solver > 0.5 (selfmask.py:231, bilateral_solver.py:185) on the device: f64 -> u8 {0,1}
__global__ __launch_bounds__(256) void threshold_f64_kernel(const double* x, double thr, unsigned char* out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] > thr ? 1 : 0;
}
extern "C" int zh_threshold_f64_u8(const double* x, double threshold, unsigned char* out, long n, hipStream_t stream) {
  ZH_CHECK_ARG(x && out && n > 0, "zh_threshold_f64_u8: bad arguments");
  hipLaunchKernelGGL(threshold_f64_kernel, dim3(zh_cdiv(n, 256)), dim3(256), 0, stream, x, threshold, out, n);
  ZH_CHECK_LAUNCH("zh_threshold_f64_u8");
  return ZH_OK;
}
