// sc_compat.hip — input staging, stage A (compat_graph) and the exclusive scan.
//
// Stage A's output is 4 N^2 bytes of weights + N^2/8 bytes of adjacency bits (103 MB at N = 5000): the HBM write
// roofline is its bound; see the stage-A section below for the tiling that gets the arithmetic out of the way.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "sc_arith.hpp"
#include "sc_block.hpp"
#include "sc_kernels.hpp"

namespace sc {

// ------------------------------------------------------------------------------------------------
// input staging: user layout -> 6 zero-padded planes, plus a finiteness check
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stage_points_kernel(const float* __restrict__ src,
                                                           const float* __restrict__ tgt, int n, int ld,
                                                           int layout, float* __restrict__ planes,
                                                           uint32_t* __restrict__ bad_flag,
                                                           uint32_t* __restrict__ zero, uint32_t zero_words,
                                                           uint32_t* __restrict__ coord_max,
                                                           uint32_t* __restrict__ coord_part,
                                                           uint32_t* __restrict__ mx_ticket,
                                                           uint64_t* __restrict__ host_max,
                                                           uint64_t* __restrict__ host_box,
                                                           uint32_t* __restrict__ zero2, uint32_t zero2_words) {
  int m = blockIdx.x * 256 + threadIdx.x;
  for (uint32_t z = (uint32_t)m; z < zero_words; z += gridDim.x * 256) zero[z] = 0u;  // the per-call control block
  for (uint32_t z = (uint32_t)m; z < zero2_words; z += gridDim.x * 256) zero2[z] = 0u;  // (stage A's deg+ accumulators)
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (m < n) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      size_t idx = layout ? ((size_t)c * n + m) : ((size_t)m * 3 + c);
      v[c] = src[idx];
      v[3 + c] = tgt[idx];
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 6; c++) ok = ok && (fabsf(v[c]) < __builtin_inff());
    if (!ok) atomicOr(bad_flag, 1u);
  }
  if (coord_max) {  // largest |coordinate| of either cloud and the two bounding boxes (C2's filters scale and centre by them)
    // coord_max: FX_MX_WORDS words — [0] max |p|, [1] max |q| (non-negative floats order as integers), [2 + c] key(max of
    // coordinate c), [8 + c] key(max of MINUS coordinate c), c = px py pz qx qy qz (float_key orders all floats as
    // unsigned integers, key 0 = "nothing yet").  Pad lanes (m >= n) contribute nothing to the boxes.
    // No same-address atomics (14 words of one cache line from every wave: 15 us): every block leaves its 14 maxima in
    // its own row of coord_part; the block that takes the last ticket reduces the rows.
    __shared__ uint32_t red[4][14];
    uint32_t k14[14];
    k14[0] = __float_as_uint(fmaxf(fabsf(v[0]), fmaxf(fabsf(v[1]), fabsf(v[2]))));
    k14[1] = __float_as_uint(fmaxf(fabsf(v[3]), fmaxf(fabsf(v[4]), fabsf(v[5]))));
#pragma unroll
    for (int c = 0; c < 6; c++) { k14[2 + c] = m < n ? float_key(v[c]) : 0u; k14[8 + c] = m < n ? float_key(-v[c]) : 0u; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int c = 0; c < 14; c++) k14[c] = max(k14[c], (uint32_t)__shfl_xor(k14[c], o));
    if ((threadIdx.x & 63) == 0)
#pragma unroll
      for (int c = 0; c < 14; c++) red[threadIdx.x >> 6][c] = k14[c];
    __syncthreads();
    if (threadIdx.x < 14)
      coord_part[(size_t)blockIdx.x * 16 + threadIdx.x] = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
    // mx_ticket == nullptr (a host-free call): no workgroup stays behind — stage A's first workgroup reduces the rows (stats_reduce_wave)
    __shared__ uint32_t s_last;
    if (mx_ticket == nullptr) {
      if (threadIdx.x == 0) s_last = 0u;
    } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const uint32_t t = __hip_atomic_fetch_add(mx_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == gridDim.x - 1) ? 1u : 0u;
      if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    }
    __syncthreads();
    if (s_last) {  // (block-uniform) the last block: reduce the rows, publish
      __shared__ uint32_t fin[16][16];
      const uint32_t c = threadIdx.x & 15, r0 = threadIdx.x >> 4;  // 16 row groups x 16 words
      uint32_t best = 0;
      if (c < 14)
        for (uint32_t r = r0; r < gridDim.x; r += 16)
          best = max(best, __hip_atomic_load(&coord_part[(size_t)r * 16 + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      fin[r0][c] = best;
      __syncthreads();
      if (threadIdx.x < 14) {
        uint32_t b2 = 0;
        for (int r = 0; r < 16; r++) b2 = max(b2, fin[r][threadIdx.x]);
        coord_max[threadIdx.x] = b2;   // (plain stores: the next kernels of the stream read them)
        fin[0][threadIdx.x] = b2;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        // the host gets the two maxima (max |q| << 32 | max |p|, bit patterns) and the boxes: it decides from them which C2
        // kernel can work at this tau (sc_capi.hip decide_filter); nothing waits for it
        if (host_box && host_max)
          for (int k = 0; k < 6; k++)  // (relaxed system-scope stores: the release of host_max below orders them)
            __hip_atomic_store(&host_box[k], ((uint64_t)fin[0][8 + k] << 32) | fin[0][2 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(mx_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
        // (host_max == nullptr: a host-free call — nothing on the host looks at these words before the call's last kernel, which
        // hands them over with the winner: launch_finalize's DeferredPub.  A system-scope release here costs the staging kernel ~1 us
        // and the launch behind it another: measured, profiles/r05_ab_deferred_publish.txt)
        if (host_max) publish_host(host_max, ((uint64_t)fin[0][1] << 32) | fin[0][0]);  // last: the host takes this word as "the boxes are there too"
      }
    }
  }
  if (m >= ld) return;
#pragma unroll
  for (int c = 0; c < 6; c++) planes[(size_t)c * ld + m] = v[c];
  // AoS copy behind the planes (8 floats per correspondence, 32-byte aligned): stage A fetches a ROW operand with one
  // s_load_dwordx8 instead of six scalar loads and their address arithmetic
  float4* aos = reinterpret_cast<float4*>(planes + 6 * (size_t)ld);
  aos[2 * (size_t)m] = make_float4(v[0], v[1], v[2], v[3]);
  aos[2 * (size_t)m + 1] = make_float4(v[4], v[5], 0.0f, 0.0f);
}

void launch_stage_points(const float* d_src, const float* d_tgt, int n, int ld, int layout, float* planes,
                         uint32_t* bad_flag, uint32_t* zero, uint32_t zero_words, uint32_t* coord_max, uint32_t* coord_part,
                         uint32_t* mx_ticket, uint64_t* host_max, uint64_t* host_box, hipStream_t st, uint32_t* zero2,
                         uint32_t zero2_words) {
  hipLaunchKernelGGL(stage_points_kernel, dim3((ld + 255) / 256), dim3(256), 0, st, d_src, d_tgt, n, ld, layout,
                     planes, bad_flag, zero, zero_words, coord_max, coord_part, mx_ticket, host_max, host_box, zero2,
                     zero2_words);
}

// ------------------------------------------------------------------------------------------------
// stage A
//
// The matrix is symmetric and the pair arithmetic is bit-symmetric (dist3(a,b) == dist3(b,a)), so only tiles on or
// above the diagonal are evaluated: one wave = one tile of TILE_R rows x 64 columns (lane = column).
//   * direct half:   S[i][j] is stored row by row (256 B contiguous per wave store); __ballot of the edge predicate
//                    IS the adjacency word of row i for this column block — lane r keeps the word of row r;
//   * mirrored half: the tile goes through an LDS transpose and S[j][i0..i0+31] is stored as 128-B row segments; the
//                    mirrored adjacency half-word of row j accumulates in lane j (bit r = edge with row i0 + r);
//   * tiles crossing the diagonal evaluate their full rectangle and write the direct half only.
// PMC on the previous one-sided kernel: WRITE_SIZE == algorithmic bytes, VALU-bound (74 VALU per pair); evaluating
// each pair once halves that work, which is what moves the kernel toward the HBM write roofline.
// deg / deg+ / word-prefix popcounts come from the bit rows in a second, tiny kernel.
// ------------------------------------------------------------------------------------------------
struct alignas(32) RowPt { float px, py, pz, qx, qy, qz, pad0, pad1; };  // the AoS copy written by stage_points_kernel

// value of `v` in lane `src` (wave-uniform src) as a scalar: v_readlane_b32 works on the bit pattern
__device__ __forceinline__ float bcast(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}


// S stores: 16 bytes per lane by default (4 consecutive floats of one row piece: a quarter of the store instructions
// and of the addresses the texture path has to process), optionally non-temporal (S is never read again on this
// path: no reason to keep it in L2 / Infinity Cache).  `mode` bit 0: 4-byte stores (round 1's form: better on small
// matrices, see launch_compat), bit 1: non-temporal.
typedef float f32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_s4(float* p, float a, float b, float c, float d, bool nt) {
  const f32x4s v = {a, b, c, d};
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4s*>(p));
  else *reinterpret_cast<f32x4s*>(p) = v;
}
__device__ __forceinline__ void store_s1(float* p, float a, bool nt) {
  if (nt) __builtin_nontemporal_store(a, p);
  else *p = a;
}

// The staging kernel's per-workgroup coordinate statistics -> the FX_MX_WORDS words of the call (one wave; the rows were written by
// the launch before this one: plain loads).  Lane = (row group lane / 16, word lane % 16).
struct StatsJob { const uint32_t* part; uint32_t rows; uint32_t* out; };
__device__ __forceinline__ void stats_reduce_wave(const StatsJob& sj, int lane) {
  const uint32_t c = (uint32_t)lane & 15u;
  uint32_t best = 0;
  if (c < 14)
    for (uint32_t r = (uint32_t)lane >> 4; r < sj.rows; r += 4) best = max(best, sj.part[(size_t)r * 16 + c]);
  best = max(best, (uint32_t)__shfl_xor((int)best, 16));
  best = max(best, (uint32_t)__shfl_xor((int)best, 32));
  if (lane < 14) sj.out[lane] = best;
}

// TILE_R rows per tile (16 or 64): 64 / TILE_R tiles stack into one 64-row block.
// DENSE: write the weight matrix S (false: SC_FLAG_NO_DENSE_S — adjacency bits only, no LDS tile image, no S traffic).
// RECT:  one-sided form for a ROW BLOCK [row0, row1) x all columns (stage A sharded by row blocks):
//        every tile writes its direct half only — the mirrored half belongs to another rank — and S holds the rows
//        of the block only (row i at S[(i - row0) * ld]).  Bit rows go to `bits` at their global row index.
template <int TILE_R, int COMPAT_WAVES, bool DENSE, bool RECT>
__global__ __launch_bounds__(64 * COMPAT_WAVES) void compat_tiles_kernel(const float* __restrict__ planes, int n,
                                                                         int ld, float d_thr, float min_len,
                                                                         float nis, float* __restrict__ S,
                                                                         uint64_t* __restrict__ bits, int n_tiles,
                                                                         int two_phase, int row0, int row1,
                                                                         int mode, uint32_t* __restrict__ degp,
                                                                         const uint32_t* __restrict__ wg_map, StatsJob sj) {
  constexpr int TILE_PAD = TILE_R + 1;  // LDS row stride of the transposed tile: conflict-free both ways
  constexpr int SUB = 64 / TILE_R;      // tiles per 64-row block
  __shared__ float tileT[DENSE ? COMPAT_WAVES : 1][DENSE ? 64 * TILE_PAD : 1];  // [column][row] per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // a host-free call's coordinate statistics: reduced HERE, by one wave of this launch, instead of by a workgroup of the staging
  // launch that stays behind for it (release + ticket + acquire: ~2.5 us of that launch); nothing before stage B's vote reads them
  if (sj.part && blockIdx.x == 0 && wave == COMPAT_WAVES - 1) stats_reduce_wave(sj, lane);
  // wg_map (whole-matrix form, COMPAT_WAVES == SUB: a workgroup = one 64 x 64 block of the matrix): which block this
  // workgroup takes — the XCD-aware order of compat_wg_map() below; ~0: a padding workgroup
  const uint32_t wg = wg_map ? wg_map[blockIdx.x] : blockIdx.x;
  if (wg == 0xFFFFFFFFu) return;
  const int t = (int)wg * COMPAT_WAVES + wave;
  if (t >= n_tiles) return;  // whole wave; no block-level barrier is used below
  const int W = ld >> 6;
  const bool nt = (mode & 2) != 0, st4 = (mode & 1) != 0;
  int J, h;
  if (RECT) {
    // column block fastest: consecutive waves write adjacent 256-byte pieces of the same rows
    J = t % W;
    h = t / W + row0 / TILE_R;  // row0 is a multiple of 64
  } else {
    // tile t -> (column block J, row sub-block h):  t = SUB J (J + 1) / 2 + h,  0 <= h < SUB (J + 1)
    J = (int)((__builtin_sqrtf(8.0f * (float)(t / SUB) + 1.0f) - 1.0f) * 0.5f);
    while (SUB * (J + 1) * (J + 2) / 2 <= t) J++;
    while (SUB * J * (J + 1) / 2 > t) J--;
    h = t - SUB * J * (J + 1) / 2;
  }
  const int rlim = RECT ? row1 : n;     // rows at or beyond it are not this launch's
  const int srow = RECT ? row0 : 0;     // S row offset
  const int i0 = h * TILE_R;
  const int j = J * 64 + lane;
  const bool diag = (h / SUB) == J;
  const float* px = planes;
  const float* py = planes + ld;
  const float* pz = planes + 2 * (size_t)ld;
  const float* qx = planes + 3 * (size_t)ld;
  const float* qy = planes + 4 * (size_t)ld;
  const float* qz = planes + 5 * (size_t)ld;
  const float jpx = px[j], jpy = py[j], jpz = pz[j], jqx = qx[j], jqy = qy[j], jqz = qz[j];
  const int i0s = __builtin_amdgcn_readfirstlane(i0);  // provably wave-uniform row base
  float* myT = tileT[DENSE ? wave : 0];
  uint64_t rowword = 0;      // lane r: adjacency word of row i0 + r over this column block
  uint64_t colword = 0;      // lane c: bit r = edge (row i0 + r, column j)
  // Interior tiles (every row and column real, not on the diagonal) are ~97 % of the work.
  const bool interior = !diag && (i0 + TILE_R <= rlim) && (J * 64 + 64 <= n);
  if (interior && two_phase) {
    // ---- two-phase form ----------------------------------------------------------------------------------
    // Only ~4 % of the pairs are edges, yet the exact chain (two correctly rounded square roots, the exponential)
    // is ~60 of the ~83 VALU instructions a pair used to cost, and the kernel was VALU-issue-bound.  Phase 1 computes
    // the two squared lengths (the same fp32 values the exact chain starts from) and a CONSERVATIVE candidate test in
    // squared quantities — no sqrt, no exp; phase 2 runs the exact chain on the queued candidates only and scatters
    // weights and adjacency bits into the LDS image of the tile, which is then written out as before.
    // Candidate test (theta = d_thr, L = min_len, A = sqrt(n2p), B = sqrt(n2q) as reals):
    //   an edge has |dp - dq| <= theta (1 + 2^-24) with dp = A (1 + e1), dq = B (1 + e2), |e| <= 2^-24, hence
    //   A^2 + B^2 - theta^2 - sigma <= 2 A B  with  sigma = (theta^2 + A^2 + B^2) 2^-21  (generous).
    //   a = fma(RN(n2p + n2q), 1 - 2^-20, -theta^2 (1 + 2^-20)) is <= that left side; so an edge has a <= 0 or
    //   a^2 <= 4 n2p n2q (1 + 2^-18).  Products so small that they could round to zero count as candidates.
    //   dp >= L needs n2p >= L^2 (1 - 2^-23); the test uses L^2 (1 - 2^-20), rounded down.
    // False positives cost a queue slot; a false negative is impossible, so the output is bit-identical.
    // Measured (r01): C2 29.4 -> 26.6 us, C3 ~400 -> ~385 us.  Ablation on C3 (N = 20 000, 1.65 GB of S): arithmetic
    // and LDS work alone 187 us, the stores alone 334 us (4.9 TB/s: 256-byte row pieces + 64-byte mirrored pieces; a
    // plain fill of the same bytes runs at 6.8 TB/s), both 383 us — the store PATTERN is the limiter now, not VALU.
    // A one-sided row kernel (1 KiB zero-fill stores per instruction + 4-byte patches of the edges, both triangles
    // evaluated) was built and was bit-exact but no faster: 234 us of arithmetic + ~190 us of stores that did not
    // overlap (423 us); 64-row tiles (256-byte mirrored pieces, SC_COMPAT_ROWS=64) change nothing on C3 either.
    constexpr int QCAP = 320;  // candidate queue per wave (a row adds <= 64; drained above QCAP - 64)
    __shared__ uint16_t queue[COMPAT_WAVES][QCAP];
    __shared__ unsigned long long rowbits[COMPAT_WAVES][TILE_R];
    __shared__ unsigned long long colbits[COMPAT_WAVES][64];
    uint16_t* myQ = queue[wave];
    unsigned long long* myRB = rowbits[wave];
    unsigned long long* myCB = colbits[wave];
    if (DENSE)
      for (int k = lane; k < 64 * TILE_PAD; k += 64) myT[k] = 0.0f;
    if (lane < TILE_R) myRB[lane] = 0ull;
    if (!RECT) myCB[lane] = 0ull;
    const RowPt* __restrict__ aos = reinterpret_cast<const RowPt*>(planes + 6 * (size_t)ld);
    // lane r (< TILE_R) keeps row point i0 + r in registers: phase 2 fetches both points of a candidate with
    // ds_bpermute (no memory latency in the drain — with ~1.5 tiles per wave slot the kernel's time is a wave's latency)
    const RowPt mine = aos[i0 + (lane & (TILE_R - 1))];
    const float theta2 = d_thr * d_thr;
    const float c1 = 1.0f - 0x1p-20f;
    const float c2 = theta2 * (1.0f + 0x1p-20f) * (1.0f + 0x1p-22f);  // rounded up a little: never below theta^2 (1 + 2^-20)
    const float k4 = 4.0f + 0x1p-16f;                                   // 4 (1 + 2^-18), exact
    const float Lc = (min_len * min_len) * (1.0f - 0x1p-20f) * (1.0f - 0x1p-22f);
    uint32_t cnt = 0;  // queued candidates (wave-uniform)
    auto drain = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // LDS is in-order within a wave
      for (uint32_t q0 = 0; q0 < cnt; q0 += 64) {  // wave-uniform trips: ds_bpermute reads 0 from inactive lanes
        const bool valid = q0 + lane < cnt;
        const uint32_t code = valid ? myQ[q0 + lane] : 0u;
        const int r = (int)(code >> 6), c = (int)(code & 63);
        const float dp = dist3_fast(__shfl(mine.px, r), __shfl(mine.py, r), __shfl(mine.pz, r), __shfl(jpx, c),
                                    __shfl(jpy, c), __shfl(jpz, c));
        const float dq = dist3_fast(__shfl(mine.qx, r), __shfl(mine.qy, r), __shfl(mine.qz, r), __shfl(jqx, c),
                                    __shfl(jqy, c), __shfl(jqz, c));
        bool edge;
        const float s = pair_weight(dp, dq, d_thr, min_len, nis, edge);
        if (valid && edge) {
          if (DENSE) myT[c * TILE_PAD + r] = s;
          atomicOr(&myRB[r], 1ull << c);
          if (!RECT) atomicOr(&myCB[c], 1ull << r);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      cnt = 0;
    };
    // The scalar unit (one per CU) was the limiter of the first two-phase form (666 SALU + 101 SMEM per wave): one
    // 32-byte scalar load per row, and the test folded into VALU selects so that ONE compare produces a lane mask.
    const float INF = __builtin_inff(), QNAN = __builtin_nanf("");
    RowPt nxt = aos[i0s];  // wave-uniform address: scalar load, issued one row ahead of its use
#pragma unroll 4
    for (int r = 0; r < TILE_R; r++) {
      const RowPt rp = nxt;
      nxt = aos[i0s + ((r + 1) & (TILE_R - 1))];
      const float dxp = rp.px - jpx, dyp = rp.py - jpy, dzp = rp.pz - jpz;
      const float dxq = rp.qx - jqx, dyq = rp.qy - jqy, dzq = rp.qz - jqz;
      const float n2p = fma_(dzp, dzp, fma_(dyp, dyp, dxp * dxp));  // exactly dist3's radicand
      const float n2q = fma_(dzq, dzq, fma_(dyq, dyq, dxq * dxq));
      const float m = fmaxf(fma_(n2p + n2q, c1, -c2), 0.0f);        // a <= 0 or a^2 <= R  <=>  max(a,0)^2 <= R
      const float t = n2p * n2q;
      const float R = (t < 0x1p-100f) ? INF : t * k4;               // vanishing products: candidate
      const float lhs = (fminf(n2p, n2q) >= Lc) ? m * m : QNAN;     // too short: NaN never compares true
      const bool cand = lhs <= R;
      const uint64_t bal = __ballot(cand);
      if (bal != 0) {  // wave-uniform
        if (cand)
          myQ[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] =
              (uint16_t)((r << 6) | lane);
        cnt += (uint32_t)__popcll(bal);
        if (cnt > (uint32_t)(QCAP - 64)) drain();
      }
    }
    if (cnt) drain();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // direct half from the tile image ([column][row], stride TILE_PAD): 16 bytes per lane — lane (r', c4) stores four
    // consecutive columns of row r' (four rows x 256 bytes per instruction; the LDS reads are conflict-free:
    // bank = 4 (lane % 16) + 17 u + lane / 16) — or, mode bit 0, one 256-byte row per instruction with 4-byte stores
    if (DENSE) {
      if (!st4) {
        const int c4 = (lane & 15) * 4, rr = lane >> 4;
#pragma unroll 4
        for (int q = 0; q < TILE_R; q += 4) {
          const int r = q + rr;
          store_s4(&S[(size_t)(i0 - srow + r) * ld + J * 64 + c4], myT[c4 * TILE_PAD + r], myT[(c4 + 1) * TILE_PAD + r],
                   myT[(c4 + 2) * TILE_PAD + r], myT[(c4 + 3) * TILE_PAD + r], nt);
        }
      } else {
#pragma unroll 4
        for (int r = 0; r < TILE_R; r++) store_s1(&S[(size_t)(i0 - srow + r) * ld + j], myT[lane * TILE_PAD + r], nt);
      }
    }
    if (lane < TILE_R) bits[(size_t)(i0 + lane) * W + J] = myRB[lane];
    // deg+ (edges to higher indices) of the tile's rows, accumulated where the adjacency words are made (degp zeroed by the
    // staging kernel): the row kernel that used to re-read the whole bit matrix for it is gone from the hot path — the
    // tiles on or above the diagonal hold exactly the upper triangle.  Integer sums: order-free.
    if (!RECT && degp && lane < TILE_R) {
      const uint32_t pc = (uint32_t)__popcll(myRB[lane]);
      if (pc) atomicAdd(&degp[i0 + lane], pc);
    }
    if (!RECT) {
      colword = myCB[lane];
      if (DENSE) {
        if (!st4) {
          // mirrored half: row J*64 + cc of S, columns i0 .. i0 + TILE_R - 1; a row piece is TILE_R / 4 lanes x 16 bytes
          constexpr int LP = TILE_R / 4, RPI = 64 / LP;  // lanes per piece, pieces (output rows) per instruction
          const int r4 = (lane % LP) * 4, co = lane / LP;
#pragma unroll 4
          for (int c = 0; c < 64; c += RPI) {
            const int cc = c + co;
            store_s4(&S[(size_t)(J * 64 + cc) * ld + i0 + r4], myT[cc * TILE_PAD + r4], myT[cc * TILE_PAD + r4 + 1],
                     myT[cc * TILE_PAD + r4 + 2], myT[cc * TILE_PAD + r4 + 3], nt);
          }
        } else {
          const int r = lane & (TILE_R - 1), hi = lane / TILE_R;
#pragma unroll 4
          for (int c = 0; c < 64; c += SUB) {
            const int cc = c + hi;
            store_s1(&S[(size_t)(J * 64 + cc) * ld + i0 + r], myT[cc * TILE_PAD + r], nt);
          }
        }
      }
      if (TILE_R == 64) bits[(size_t)j * W + (i0 >> 6)] = colword;
      else if (TILE_R == 32) reinterpret_cast<uint32_t*>(bits)[((size_t)j * W + (i0 >> 6)) * 2 + (h % SUB)] = (uint32_t)colword;
      else reinterpret_cast<uint16_t*>(bits)[((size_t)j * W + (i0 >> 6)) * 4 + (h % SUB)] = (uint16_t)colword;
    }
    return;
  }
  auto row_pair = [&](int r, auto guarded) {
    constexpr bool G = decltype(guarded)::value;
    float dpa, dqa, dpb, dqb;
    {  // row operands: wave-uniform addresses -> scalar loads, no VALU work
      const float ipx = px[i0s + r], ipy = py[i0s + r], ipz = pz[i0s + r], iqx = qx[i0s + r], iqy = qy[i0s + r],
                  iqz = qz[i0s + r];
      dpa = dist3_fast(ipx, ipy, ipz, jpx, jpy, jpz);
      dqa = dist3_fast(iqx, iqy, iqz, jqx, jqy, jqz);
    }
    {
      const float ipx = px[i0s + r + 1], ipy = py[i0s + r + 1], ipz = pz[i0s + r + 1], iqx = qx[i0s + r + 1],
                  iqy = qy[i0s + r + 1], iqz = qz[i0s + r + 1];
      dpb = dist3_fast(ipx, ipy, ipz, jpx, jpy, jpz);
      dqb = dist3_fast(iqx, iqy, iqz, jqx, jqy, jqz);
    }
    const int ia = i0 + r, ib = ia + 1;
    const float da = fabsf(dpa - dqa), db = fabsf(dpb - dqb);
    bool ea = (da <= d_thr) && (dpa >= min_len) && (dqa >= min_len);
    bool eb = (db <= d_thr) && (dpb >= min_len) && (dqb >= min_len);
    if (G) {
      ea = ea && (j != ia) && (j < n) && (ia < rlim);
      eb = eb && (j != ib) && (j < n) && (ib < rlim);
    }
    const uint64_t worda = __ballot(ea), wordb = __ballot(eb);
    float sa = 0.0f, sb = 0.0f;
    if ((worda | wordb) != 0) {
      const float va = sc_expf((da * da) * nis), vb = sc_expf((db * db) * nis);
      sa = ea ? va : 0.0f;
      sb = eb ? vb : 0.0f;
    }
    if (DENSE) {
      if (!G || ia < rlim) store_s1(&S[(size_t)(ia - srow) * ld + j], sa, nt);
      if (!G || ib < rlim) store_s1(&S[(size_t)(ib - srow) * ld + j], sb, nt);
    }
    if (lane == r) rowword = worda;
    if (lane == r + 1) rowword = wordb;
    if (!RECT) {
      colword |= (ea ? (1ull << r) : 0ull) | (eb ? (2ull << r) : 0ull);
      if (DENSE && !diag) { myT[lane * TILE_PAD + r] = sa; myT[lane * TILE_PAD + r + 1] = sb; }
    }
  };
  if (interior) {
#pragma unroll 2
    for (int r = 0; r < TILE_R; r += 2) row_pair(r, std::false_type{});
  } else {
#pragma unroll 2
    for (int r = 0; r < TILE_R; r += 2) row_pair(r, std::true_type{});
  }
  if (lane < TILE_R && i0 + lane < rlim) bits[(size_t)(i0 + lane) * W + J] = rowword;
  if (!RECT && degp && lane < TILE_R && i0 + lane < rlim) {  // (a tile on the diagonal: only the columns beyond the row itself)
    const int b = (i0 + lane) & 63;
    const uint64_t up = diag ? (b == 63 ? 0ull : (rowword & (~0ull << (b + 1)))) : rowword;
    const uint32_t pc = (uint32_t)__popcll(up);
    if (pc) atomicAdd(&degp[i0 + lane], pc);
  }
  if (!RECT && !diag) {
    // mirrored half: row j of S, columns i0 .. i0 + TILE_R - 1; SUB output rows per instruction
    if (DENSE) {
      const int r = lane & (TILE_R - 1), hi = lane / TILE_R;
#pragma unroll 4
      for (int c = 0; c < 64; c += SUB) {
        const int cc = c + hi;
        const float v = myT[cc * TILE_PAD + r];
        const int jj = J * 64 + cc;
        if (jj < n) store_s1(&S[(size_t)jj * ld + i0 + r], v, nt);
      }
    }
    // mirrored adjacency: row j, piece (h % SUB) of word (i0 / 64)
    if (j < n) {
      if (TILE_R == 64) bits[(size_t)j * W + (i0 >> 6)] = colword;
      else if (TILE_R == 32) reinterpret_cast<uint32_t*>(bits)[((size_t)j * W + (i0 >> 6)) * 2 + (h % SUB)] = (uint32_t)colword;
      else reinterpret_cast<uint16_t*>(bits)[((size_t)j * W + (i0 >> 6)) * 4 + (h % SUB)] = (uint16_t)colword;
    }
  }
}

// deg[i], degp[i] (bits above i) and wpre[i][w] = #set bits of row i in words [0, w): one wave per row.
// rowcost (optional): estimate of stage B's work for the edges (i, j), j > i, of row i — the words the counting pass
// ANDs for each of them plus a constant: sum over j of (W - j / 64 + 4), saturated to u32.  Its prefix over the rows
// splits the rows into contiguous, equally heavy ranges when stage B is sharded.
__global__ __launch_bounds__(256) void row_stats_kernel(const uint64_t* __restrict__ bits, int n, int W,
                                                        uint32_t* __restrict__ deg, uint32_t* __restrict__ degp,
                                                        uint32_t* __restrict__ wpre,
                                                        uint64_t* __restrict__ zero_rows,
                                                        uint32_t* __restrict__ rowcost) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  if (zero_rows)
    for (int w = lane; w < W; w += 64) zero_rows[(size_t)i * W + w] = 0ull;
  uint32_t d_all = 0, d_up = 0;
  uint64_t cost = 0;
  for (int wb = 0; wb < W; wb += 64) {
    const int w = wb + lane;
    const uint64_t v = w < W ? bits[(size_t)i * W + w] : 0ull;
    const uint32_t pc = (uint32_t)__popcll(v);
    uint32_t inc = pc;  // inclusive wave scan of the word popcounts
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (w < W) {
      wpre[(size_t)i * W + w] = d_all + inc - pc;
      uint64_t up = v;
      if (w < (i >> 6)) up = 0;
      else if (w == (i >> 6)) up &= ((i & 63) == 63) ? 0ull : (~0ull << ((i & 63) + 1));
      d_up += __popcll(up);
      cost += (uint64_t)__popcll(up) * (uint64_t)(W - w + 4);
    }
    d_all += __shfl(inc, 63);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d_up += __shfl_xor(d_up, o);
  if (rowcost) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cost += __shfl_xor(cost, o);
  }
  if (lane == 0) {
    deg[i] = d_all; degp[i] = d_up;
    if (rowcost) rowcost[i] = cost > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cost;
  }
}

// row_stats + the prefix over the rows in ONE launch ("fold row_stats ..."): a workgroup takes RS_ROWS rows
// (16 waves x RS_RPW rows), computes what row_stats_kernel computes, scans its edge counts (and row costs) in one wave and
// gets the sums of the tiles before it by decoupled look-back (sc_block.hpp) — so edge_off (the CSR row offsets), the
// per-row CSR bases, the cost prefix and the edge count for the host all come out of the kernel that read the bit rows;
// the separate scan launch(es) and their extra pass over deg / deg+ disappear.  157 tiles at N = 5000, 625 at 20 000 (64-row tiles: 14.7 us on C2, fewer CUs busy).
constexpr int RS_ROWS = 32;           // rows per tile: 16 waves x RS_RPW rows
constexpr int RS_RPW = RS_ROWS / 16;  // rows a wave takes, one after the other
__global__ __launch_bounds__(1024) void row_stats_scan_kernel(const uint64_t* __restrict__ bits, int n, int W,
                                                             uint32_t* __restrict__ deg, uint32_t* __restrict__ degp,
                                                             uint32_t* __restrict__ wpre,
                                                             uint64_t* __restrict__ zero_rows,
                                                             uint64_t* __restrict__ edge_off,
                                                             uint32_t* __restrict__ ebase,
                                                             uint64_t* __restrict__ cost_pre, LbArgs lb,
                                                             uint64_t* __restrict__ host_total) {
  __shared__ uint32_t l_degp[RS_ROWS], l_dlow[RS_ROWS];
  __shared__ uint64_t l_cost[RS_ROWS];
  __shared__ uint32_t s_tile;
  uint32_t* ticket = lb.ticket;
  const uint32_t epoch = lb.epoch;
  if (threadIdx.x == 0) s_tile = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const uint32_t tile = s_tile, nb = gridDim.x;
  if (tile >= nb) return;  // (a ticket that was not zero at launch: never index memory with it)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = 0; rr < RS_RPW; rr++) {
    const int slot = wave * RS_RPW + rr;
    const int i = (int)tile * RS_ROWS + slot;
    uint32_t d_all = 0, d_up = 0;
    uint64_t cost = 0;
    if (i < n) {  // wave-uniform
      if (zero_rows)
        for (int w = lane; w < W; w += 64) zero_rows[(size_t)i * W + w] = 0ull;
      for (int wb = 0; wb < W; wb += 64) {
        const int w = wb + lane;
        const uint64_t v = w < W ? bits[(size_t)i * W + w] : 0ull;
        const uint32_t pc = (uint32_t)__popcll(v);
        uint32_t inc = pc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t t = __shfl_up(inc, o);
          if (lane >= o) inc += t;
        }
        if (w < W) {
          wpre[(size_t)i * W + w] = d_all + inc - pc;
          uint64_t up = v;
          if (w < (i >> 6)) up = 0;
          else if (w == (i >> 6)) up &= ((i & 63) == 63) ? 0ull : (~0ull << ((i & 63) + 1));
          d_up += __popcll(up);
          cost += (uint64_t)__popcll(up) * (uint64_t)(W - w + 4);
        }
        d_all += __shfl(inc, 63);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { d_up += __shfl_xor(d_up, o); cost += __shfl_xor(cost, o); }
    }
    if (lane == 0) {
      if (i < n) { deg[i] = d_all; degp[i] = d_up; }
      l_degp[slot] = d_up; l_dlow[slot] = d_all - d_up; l_cost[slot] = cost;
    }
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;  // wave 0: scan of the tile's 64 rows, look-back, outputs
  const uint64_t mine_e = lane < RS_ROWS ? l_degp[lane] : 0u, mine_c = lane < RS_ROWS ? l_cost[lane] : 0ull;
  uint64_t inc_e = mine_e, inc_c = mine_c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t te = __shfl_up(inc_e, o), tc = __shfl_up(inc_c, o);
    if (lane >= o) { inc_e += te; inc_c += tc; }
  }
  const uint64_t tot_e = __shfl(inc_e, 63), tot_c = __shfl(inc_c, 63);
  uint64_t* const desc[2] = {lb.desc, lb.desc + nb};
  const uint64_t own[2] = {tot_e, tot_c};
  uint64_t pre[2];
  lb_lookback<2>(desc, tile, epoch, own, pre, lb.err);
  const int i = (int)tile * RS_ROWS + lane;
  if (lane < RS_ROWS && i < n) {
    const uint64_t off = pre[0] + inc_e - mine_e;
    edge_off[i] = off;
    ebase[i] = (uint32_t)off - l_dlow[lane];  // (lane < RS_ROWS here)
    if (cost_pre) cost_pre[i] = pre[1] + inc_c - mine_c;
  }
  if (tile == nb - 1 && lane == 0) {
    edge_off[n] = pre[0] + tot_e;
    if (cost_pre) cost_pre[n] = pre[1] + tot_c;
    if (host_total) publish_host(host_total, pre[0] + tot_e);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every tile has taken its ticket
  }
}

size_t row_stats_scan_state_bytes(int n) { return (size_t)(2 * ((n + RS_ROWS - 1) / RS_ROWS)) * sizeof(uint64_t); }

void launch_row_stats_scan(const Points& pts, const uint64_t* bits, uint32_t* deg, uint32_t* degp, uint32_t* wpre,
                           uint64_t* zero_rows, uint64_t* edge_off, uint32_t* ebase, uint64_t* cost_pre, const LbArgs& lb,
                           uint64_t* host_total, hipStream_t st) {
  hipLaunchKernelGGL(row_stats_scan_kernel, dim3((pts.n + RS_ROWS - 1) / RS_ROWS), dim3(1024), 0, st, bits, pts.n,
                     pts.ld >> 6, deg, degp, wpre, zero_rows, edge_off, ebase, cost_pre, lb, host_total);
}

// Contiguous row range of one rank: rows [lo, hi) with lo = first row whose cost prefix reaches rank / world of the
// total (row 0 for rank 0, n for the end of the last rank).  own_row[0..1] and the CSR edge range own_edge[0..1] of
// those rows go to the control block; every rank computes the same boundaries from the same replicated arrays.
__global__ __launch_bounds__(64) void shard_split_kernel(const uint64_t* __restrict__ cost_pre,
                                                         const uint64_t* __restrict__ edge_off, int n, uint32_t rank,
                                                         uint32_t world, uint32_t* __restrict__ own_row,
                                                         uint64_t* __restrict__ own_edge) {
  // one wave; both boundaries (lo = boundary `rank`, hi = boundary `rank + 1`) by a 64-ary search: every step the 64
  // lanes probe 64 evenly spaced rows at once (3 steps for n = 20 000 instead of 15 dependent loads)
  const int lane = threadIdx.x;
  const uint64_t total = cost_pre[n];
  for (int which = 0; which < 2; which++) {
    const uint32_t l = rank + (uint32_t)which;
    int row;
    if (l == 0) row = 0;
    else if (l >= world) row = n;
    else {
      const uint64_t target = (uint64_t)(((unsigned __int128)total * l) / world);
      int lo = 0, hi = n;  // invariant: the answer (first r in [0, n] with cost_pre[r] >= target) lies in [lo, hi]
      while (hi - lo > 0) {
        const int span = hi - lo, step = (span + 63) / 64;
        const int probe = lo + lane * step;                   // probes lo, lo + step, ... (those < hi are real)
        const bool ge = probe < hi ? (cost_pre[probe] >= target) : true;
        const uint64_t m = __ballot(ge);
        // first probe at or above the target; none (the last real probe is still below it): the answer lies above it
        const int first = m ? __builtin_ctzll(m) : 64;
        if (first == 0) { hi = lo; }                                                // cost_pre[lo] >= target
        else if (first == 64) { lo = lo + 63 * step + 1; }                          // in (probe[63], hi]
        else { const int plo = lo + (first - 1) * step + 1; hi = min(hi, lo + first * step); lo = plo; }  // in (probe[first-1], probe[first]]
      }
      row = lo;
    }
    if (lane == 0) { own_row[which] = (uint32_t)row; own_edge[which] = edge_off[row]; }
  }
}

// rows [row0, row1) of the graph (row0 a multiple of 64, row1 <= n); the whole matrix (symmetric tiles, each pair
// evaluated once) when the range is [0, n).  S == nullptr: adjacency bits only (SC_FLAG_NO_DENSE_S).  For a proper
// sub-range S holds the rows of the range only (row i at S + (i - row0) * ld).
// XCD-aware order of the 64 x 64 blocks (m, J), m <= J, of the symmetric form.  A 64-byte line of a bit row holds 8 adjacency
// words, and they are written by 8 DIFFERENT workgroups: row block m's words J = 8 k .. 8 k + 7 by the direct halves of (m, 8 k ..),
// and — the mirrored half — column block J's words m = 8 k' .. by (8 k' .., J).  Workgroups go to the 8 XCDs round-robin by
// index, each XCD has its own L2, and a line assembled in eight L2s leaves each of them as a partial write: the bits-only
// form of the kernel wrote 12.2 MB for 3.2 MB of bit rows (profiles/r03_pmc_compat_writes.txt).  So the blocks are dealt to
// the XCDs by 8 x 8 SUPER-blocks (all eight writers of a line in one super-block, hence one L2, and close together in its
// dispatch order): entry b of the map is the block workgroup b takes — workgroup b runs on XCD b % 8 —, ~0 pads the XCDs'
// lists to one length.  Super-blocks are dealt largest first to the XCD with the least work so far.
std::vector<uint32_t> compat_wg_map(int W) {
  const int SB = (W + 7) / 8;
  struct Super { int sj, sm; int tiles; };
  std::vector<Super> sup;
  for (int sj = 0; sj < SB; sj++)
    for (int sm = 0; sm <= sj; sm++) {
      int tiles = 0;
      for (int J = sj * 8; J < W && J < sj * 8 + 8; J++)
        for (int m = sm * 8; m <= J && m < sm * 8 + 8; m++) tiles++;
      if (tiles) sup.push_back(Super{sj, sm, tiles});
    }
  std::stable_sort(sup.begin(), sup.end(), [](const Super& a, const Super& b) { return a.tiles > b.tiles; });
  std::vector<uint32_t> lists[8];
  for (const Super& su : sup) {
    int best = 0;
    for (int x = 1; x < 8; x++) if (lists[x].size() < lists[best].size()) best = x;
    for (int J = su.sj * 8; J < W && J < su.sj * 8 + 8; J++)
      for (int m = su.sm * 8; m <= J && m < su.sm * 8 + 8; m++) lists[best].push_back((uint32_t)(J * (J + 1) / 2 + m));
  }
  size_t len = 0;
  for (int x = 0; x < 8; x++) len = lists[x].size() > len ? lists[x].size() : len;
  std::vector<uint32_t> map(8 * len, 0xFFFFFFFFu);
  for (int x = 0; x < 8; x++)
    for (size_t k = 0; k < lists[x].size(); k++) map[8 * k + x] = lists[x][k];
  return map;
}

void launch_compat(const Points& pts, const Derived& dv, float* S, uint64_t* bits, int row0, int row1, const Tuning& tn,
                   hipStream_t st, uint32_t* degp, const uint32_t* wg_map, uint32_t wg_map_len, const uint32_t* stat_part,
                   uint32_t* stat_out) {
  const StatsJob sj{stat_part, stat_part ? (uint32_t)((pts.ld + 255) / 256) : 0u, stat_out};
  const int W = pts.ld >> 6;
  const int two_phase = tn.compat_one_phase ? 0 : 1;  // the one-phase interior form stays for A/B and parity
  const bool rect = !(row0 == 0 && row1 >= pts.n);
  // S stores, measured r02 (profiles/r02_ab_compat_stores.txt): on the first box 16 bytes per lane paid on the big matrix
  // (C3, N = 20 000: 346 vs 404 us, alternating with the form four times in one process) and cost a little on the small
  // one, where the kernel's time is a wave's latency (C2: 25.8 vs 23.7 us); on another box, later, C3 ran 400-420 us in
  // EVERY form.  So: 16-byte stores from ~N = 8000 on (never slower there), 4-byte below; non-temporal stores changed
  // nothing anywhere.  Tuning::compat_store_mode forces a form (bit 0: 4-byte, bit 2: 16-byte), bit 1 adds the nt hint.
  const long long tiles_all = rect ? (long long)((row1 - row0 + 15) / 16) * W : 2ll * W * (W + 1);
  int mode = (tiles_all >= 32768 || rect) ? 0 : 1;
  if (tn.compat_store_mode & 1u) mode = 1;
  if (tn.compat_store_mode & 4u) mode = 0;
  mode |= (int)(tn.compat_store_mode & 2u);
#define SC_COMPAT_ARGS pts.planes, pts.n, pts.ld, dv.d_thr, dv.min_len, dv.neg_inv2sig2, S, bits, n_tiles, two_phase, row0, row1, mode, degp, map_arg, sj
  const uint32_t* map_arg = nullptr;  // (only the symmetric 16- and 32-row forms take the map: there a workgroup is a 64 x 64 block)
  if (rect) {
    if (row1 <= row0) return;
    // one-sided 16-row tiles over the rectangle (every pair of the block evaluated by this rank)
    const int n_tiles = ((row1 - row0 + 15) / 16) * W;
    if (S) hipLaunchKernelGGL((compat_tiles_kernel<16, 4, true, true>), dim3((n_tiles + 3) / 4), dim3(256), 0, st, SC_COMPAT_ARGS);
    else hipLaunchKernelGGL((compat_tiles_kernel<16, 4, false, true>), dim3((n_tiles + 3) / 4), dim3(256), 0, st, SC_COMPAT_ARGS);
    return;
  }
  // tile height: 16 rows (4.3 KiB LDS image per wave, 4 waves per workgroup), 32 rows (8.4 KiB, 2 waves per workgroup) or
  // 64 rows (16.6 KiB, one wave per workgroup; the mirrored half is then written as full 256-byte segments).  Measured: C2 28 vs 49 us, C3 432 vs
  // 435 us — bigger mirrored pieces do not pay for the lost occupancy.  Tuning::compat_rows = 64 selects it (experiments).
  // r02, by size (HIP-event bracket around the kernel, alternating in one process): N = 6000 31 vs 40 us, 8000 52 vs 58,
  // 10 000 97 vs 91, 12 000 137 vs 129, 16 000 264 vs 226, 20 000 (C3) 430 vs 372 for 16- vs 32-row tiles: from ~10 000
  // correspondences on the launch has enough tiles that the taller tile's lower occupancy no longer shows, and its
  // 128-byte mirrored pieces and half as many row operands per pair do.
  const int tr = tn.compat_rows == 64 ? 64 : (tn.compat_rows == 32 ? 32 : (tn.compat_rows == 16 ? 16 : (pts.n >= 10000 ? 32 : 16)));
  if (tr == 32) {
    const int n_tiles = 2 * W * (W + 1) / 2;
    map_arg = wg_map;
    const unsigned grid = wg_map ? wg_map_len : (unsigned)((n_tiles + 1) / 2);
    if (S) hipLaunchKernelGGL((compat_tiles_kernel<32, 2, true, false>), dim3(grid), dim3(128), 0, st, SC_COMPAT_ARGS);
    else hipLaunchKernelGGL((compat_tiles_kernel<32, 2, false, false>), dim3(grid), dim3(128), 0, st, SC_COMPAT_ARGS);
  } else if (tr == 64) {
    const int n_tiles = W * (W + 1) / 2;
    if (S) hipLaunchKernelGGL((compat_tiles_kernel<64, 1, true, false>), dim3(n_tiles), dim3(64), 0, st, SC_COMPAT_ARGS);
    else hipLaunchKernelGGL((compat_tiles_kernel<64, 1, false, false>), dim3(n_tiles), dim3(64), 0, st, SC_COMPAT_ARGS);
  } else {
    const int n_tiles = 4 * W * (W + 1) / 2;
    map_arg = wg_map;
    const unsigned grid = wg_map ? wg_map_len : (unsigned)((n_tiles + 3) / 4);
    if (S) hipLaunchKernelGGL((compat_tiles_kernel<16, 4, true, false>), dim3(grid), dim3(256), 0, st, SC_COMPAT_ARGS);
    else hipLaunchKernelGGL((compat_tiles_kernel<16, 4, false, false>), dim3(grid), dim3(256), 0, st, SC_COMPAT_ARGS);
  }
#undef SC_COMPAT_ARGS
}

void launch_row_stats(const Points& pts, const uint64_t* bits, uint32_t* deg, uint32_t* degp, uint32_t* wpre,
                      uint64_t* zero_rows, uint32_t* rowcost, hipStream_t st) {
  hipLaunchKernelGGL(row_stats_kernel, dim3((pts.n + 3) / 4), dim3(256), 0, st, bits, pts.n, pts.ld >> 6, deg, degp,
                     wpre, zero_rows, rowcost);
}

void launch_shard_split(const uint64_t* cost_pre, const uint64_t* edge_off, int n, uint32_t rank, uint32_t world,
                        uint32_t* own_row, uint64_t* own_edge, hipStream_t st) {
  hipLaunchKernelGGL(shard_split_kernel, dim3(1), dim3(64), 0, st, cost_pre, edge_off, n, rank, world, own_row, own_edge);
}

// ------------------------------------------------------------------------------------------------
// exclusive scan u32 -> u64: block sums, then a down-sweep whose blocks add up the sums before them (two launches);
// beyond SCAN_SELF_MAX tiles a one-block scan of the sums runs in between
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// range (optional, device): [lo, hi) outside which every input is known to be zero (sharded stage B: the per-edge
// triangle counts of the edges other ranks enumerate) — tiles wholly outside are neither read nor written.
// ... and inside a tile that straddles an end of the range the elements outside it read as zero WITHOUT being read (r05: a host-free
// call's per-edge counts beyond its real edge count are never written — sc_capi.hip, ControlBlock::live_edges)
__device__ __forceinline__ bool scan_live(const uint64_t* __restrict__ range, size_t idx) {
  return !range || ((uint64_t)idx >= range[0] && (uint64_t)idx < range[1]);
}
__device__ __forceinline__ bool scan_tile_dead(const uint64_t* __restrict__ range, size_t tile) {
  if (!range) return false;
  const uint64_t t0 = (uint64_t)tile * SCAN_TILE;
  return t0 + SCAN_TILE <= range[0] || t0 >= range[1];
}

// ---- a tile of the scans, coalesced (r05).  By time stamps inside the look-back kernel's tiles a tile's life was 3 us of loads + local scan
// and 3 us of stores around 1 us of look-back: every thread owned SCAN_ITEMS CONSECUTIVE elements, so a wave's load touched 64 pieces 64
// bytes apart and its 8-byte stores 64 lines each, sixteen times over.  Now a tile is SCAN_SUB sub-tiles of 1024 elements and a thread takes
// four consecutive elements of each: a wave reads 1 KiB and writes 2 KiB at a stretch (C2: scan_lookback_kernel 11.7 -> 9.3 us).  A piece
// inside the array is loaded BEFORE `range` is looked at (its two words were a dependent round trip at the head of the chain) and masked
// afterwards: what lies beyond a host-free call's real edge count was never written, but it is the array's own memory.
constexpr int SCAN_SUB = SCAN_ITEMS / 4;
static_assert(SCAN_ITEMS % 4 == 0 && SCAN_THREADS == 256, "sub-tiles of 4 x 256 elements, four waves");
struct ScanTile { uint32_t v[SCAN_SUB][4]; uint64_t ex[SCAN_SUB]; uint64_t tot; bool dead; };
// loads + masks the tile's elements; returns with ex[r] = sum of the tile's elements before this thread's piece of sub-tile r and tot =
// the tile's sum.  wsum: 4 x SCAN_SUB words of LDS; one barrier; every thread of the workgroup calls it.
// (mask_dead: a tile wholly outside `range` reads as zeros — t.dead says so; `range` is looked at only AFTER the loads are on their way)
__device__ __forceinline__ void scan_tile_load(const uint32_t* __restrict__ in, size_t n, size_t tile, const uint64_t* __restrict__ range,
                                               bool mask_dead, uint64_t (*wsum)[SCAN_SUB], ScanTile& t) {
  const size_t tbase = tile * SCAN_TILE + (size_t)threadIdx.x * 4;
  if ((tile + 1) * SCAN_TILE <= n) {  // (block-uniform: ONE branch around all the loads — a branch per piece made the compiler wait for each)
    uint4 q[SCAN_SUB];
#pragma unroll
    for (int r = 0; r < SCAN_SUB; r++) q[r] = *reinterpret_cast<const uint4*>(in + tbase + (size_t)r * 1024);
#pragma unroll
    for (int r = 0; r < SCAN_SUB; r++) { t.v[r][0] = q[r].x; t.v[r][1] = q[r].y; t.v[r][2] = q[r].z; t.v[r][3] = q[r].w; }
  } else {
#pragma unroll
    for (int r = 0; r < SCAN_SUB; r++)
#pragma unroll
      for (int k = 0; k < 4; k++) { const size_t idx = tbase + (size_t)r * 1024 + k; t.v[r][k] = idx < n ? in[idx] : 0u; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool dead = mask_dead && scan_tile_dead(range, tile);  // block-uniform
  t.dead = dead;
  uint64_t inc[SCAN_SUB], mine[SCAN_SUB];
#pragma unroll
  for (int r = 0; r < SCAN_SUB; r++) {
    const size_t idx = tbase + (size_t)r * 1024;
    uint64_t a = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { t.v[r][k] = (!dead && scan_live(range, idx + k)) ? t.v[r][k] : 0u; a += t.v[r][k]; }
    mine[r] = a; inc[r] = a;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
#pragma unroll
    for (int r = 0; r < SCAN_SUB; r++) {
      const uint64_t x = __shfl_up(inc[r], o);
      if (lane >= o) inc[r] += x;
    }
  if (lane == 63)
#pragma unroll
    for (int r = 0; r < SCAN_SUB; r++) wsum[wave][r] = inc[r];
  __syncthreads();
  uint64_t tot = 0;
#pragma unroll
  for (int r = 0; r < SCAN_SUB; r++) {
    uint64_t before = tot;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint64_t x = wsum[w][r]; before += w < wave ? x : 0ull; tot += x; }
    t.ex[r] = before + inc[r] - mine[r];
  }
  t.tot = tot;
}
struct ScanEbase { const uint32_t* deg; const uint32_t* degp; uint32_t* ebase; };
// out[i] = pre + (exclusive prefix inside the tile) for the tile's elements below n; eb: see scan_downsweep_kernel
__device__ __forceinline__ void scan_tile_store(uint64_t* __restrict__ out, size_t n, size_t tile, uint64_t pre, const ScanTile& t, const ScanEbase& eb) {
  const size_t tbase = tile * SCAN_TILE + (size_t)threadIdx.x * 4;
  const bool whole = (tile + 1) * SCAN_TILE <= n;  // block-uniform
#pragma unroll
  for (int r = 0; r < SCAN_SUB; r++) {
    const size_t idx = tbase + (size_t)r * 1024;
    uint64_t o4[4], run = pre + t.ex[r];
#pragma unroll
    for (int k = 0; k < 4; k++) { o4[k] = run; run += t.v[r][k]; }
    if (whole) {
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      u64x2* dst = reinterpret_cast<u64x2*>(out + idx);  // (idx is a multiple of 4: 32-byte aligned)
      dst[0] = u64x2{o4[0], o4[1]}; dst[1] = u64x2{o4[2], o4[3]};
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) if (idx + k < n) out[idx + k] = o4[k];
    }
    if (eb.ebase)
#pragma unroll
      for (int k = 0; k < 4; k++) if (idx + k < n) eb.ebase[idx + k] = (uint32_t)o4[k] - (eb.deg[idx + k] - eb.degp[idx + k]);
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                       uint64_t* __restrict__ bsum,
                                                                       const uint64_t* __restrict__ range) {
  __shared__ uint64_t wsum[4][SCAN_SUB];
  if (scan_tile_dead(range, blockIdx.x)) { if (threadIdx.x == 0) bsum[blockIdx.x] = 0; return; }
  ScanTile t;
  scan_tile_load(in, n, blockIdx.x, range, false, wsum, t);
  if (threadIdx.x == 0) bsum[blockIdx.x] = t.tot;
}

__global__ __launch_bounds__(1024) void scan_of_sums_kernel(uint64_t* __restrict__ bsum, size_t nb,
                                                            uint64_t* __restrict__ total_out,
                                                            uint64_t* __restrict__ host_total) {
  __shared__ uint64_t lds[16];
  uint64_t carry = 0;
  for (size_t b0 = 0; b0 < nb; b0 += 1024) {
    const size_t b = b0 + threadIdx.x;
    const uint64_t v = b < nb ? bsum[b] : 0;
    uint64_t tot;
    const uint64_t ex = block_exscan_u64(v, lds, &tot);
    if (b < nb) bsum[b] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    *total_out = carry;
    if (host_total) publish_host(host_total, carry);
  }
}

// SELF: every block sums the raw block sums before it by itself (<= SCAN_SELF_MAX of them, from L2) and the last one
// writes the total — the single-block scan-of-sums launch (a ~4.6 us floor) disappears.
// eb (optional): also ebase[i] = (u32) out[i] - (deg[i] - degp[i]), the CSR base of row i (see launch_edge_fill)
template <bool SELF>
__global__ __launch_bounds__(SCAN_THREADS) void scan_downsweep_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                      const uint64_t* __restrict__ bsum,
                                                                      uint64_t* __restrict__ out,
                                                                      uint64_t* __restrict__ host_total,
                                                                      const uint64_t* __restrict__ range,
                                                                      ScanEbase eb) {
  __shared__ uint64_t lds[8];
  __shared__ uint64_t wsum[4][SCAN_SUB];
  const bool dead = scan_tile_dead(range, blockIdx.x) && !(SELF && blockIdx.x == gridDim.x - 1);  // block-uniform
  if (dead) return;
  ScanTile t;
  scan_tile_load(in, n, blockIdx.x, range, false, wsum, t);
  uint64_t pre;
  if (SELF) {
    uint64_t a = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += SCAN_THREADS) a += bsum[b];
    pre = block_reduce_u64(a, lds);
  } else {
    pre = bsum[blockIdx.x];
  }
  scan_tile_store(out, n, blockIdx.x, pre, t, eb);
  if (SELF && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    out[n] = pre + t.tot;
    if (host_total) publish_host(host_total, pre + t.tot);
  }
}

// Single-pass form (decoupled look-back, sc_block.hpp): one launch instead of block sums + down-sweep.
// lb: ticket (zero between launches: the last tile resets it), error flag, one descriptor per tile, this launch's epoch.
__global__ __launch_bounds__(SCAN_THREADS) void scan_lookback_kernel(const uint32_t* __restrict__ in, size_t n,
                                                                     uint64_t* __restrict__ out, LbArgs lb,
                                                                     uint64_t* __restrict__ host_total,
                                                                     const uint64_t* __restrict__ range, ScanEbase eb) {
  __shared__ uint32_t s_tile;
  __shared__ uint64_t s_prefix;
  uint32_t* ticket = lb.ticket;
  const uint32_t epoch = lb.epoch;
  if (threadIdx.x == 0) s_tile = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const uint32_t tile = s_tile;
  if (tile >= gridDim.x) return;  // (a ticket that was not zero at launch: never index memory with it)
  __shared__ uint64_t wsum[4][SCAN_SUB];
  ScanTile t;
  scan_tile_load(in, n, tile, range, true, wsum, t);  // (a tile wholly outside the range: known zeros, never written)
  const bool dead = t.dead;
  const uint64_t tot = t.tot;
  if (threadIdx.x < 64) {
    uint64_t* const desc[1] = {lb.desc};
    const uint64_t own[1] = {tot};
    uint64_t pre[1];
    lb_lookback<1>(desc, tile, epoch, own, pre, lb.err);
    if (threadIdx.x == 0) s_prefix = pre[0];
  }
  __syncthreads();
  const uint64_t pre = s_prefix;
  if (!dead) scan_tile_store(out, n, tile, pre, t, eb);
  if (tile == gridDim.x - 1 && threadIdx.x == 0) {
    out[n] = pre + tot;
    if (host_total) publish_host(host_total, pre + tot);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every tile has taken its ticket
  }
}

// small inputs: ONE block scans up to two arrays in one launch (three launches of the tiled scan are pure latency there)
__global__ __launch_bounds__(1024) void scan_small_kernel(const uint32_t* __restrict__ in0, uint64_t* __restrict__ out0,
                                                          const uint32_t* __restrict__ in1, uint64_t* __restrict__ out1,
                                                          size_t n, uint64_t* __restrict__ host_total) {
  __shared__ uint64_t lds[16];
  constexpr int PER = 8;  // consecutive elements per thread and pass: 8192 per pass, so a few passes at most
  const int arrays = in1 ? 2 : 1;
  for (int a = 0; a < arrays; a++) {
    const uint32_t* in = a ? in1 : in0;
    uint64_t* out = a ? out1 : out0;
    uint64_t carry = 0;
    for (size_t b0 = 0; b0 < n; b0 += 1024 * PER) {
      const size_t base = b0 + (size_t)threadIdx.x * PER;
      uint32_t v[PER];
      uint64_t sum = 0;
#pragma unroll
      for (int k = 0; k < PER; k++) { v[k] = (base + k < n) ? in[base + k] : 0u; sum += v[k]; }
      uint64_t tot;
      uint64_t run = carry + block_exscan_u64(sum, lds, &tot);
#pragma unroll
      for (int k = 0; k < PER; k++) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
      }
      carry += tot;
    }
    if (threadIdx.x == 0) {
      out[n] = carry;
      if (a == 0 && host_total) publish_host(host_total, carry);
    }
  }
}

constexpr size_t SCAN_SMALL_MAX = 8192;  // one pass of the single block; beyond it the tiled form is faster (N = 20 000: 26 -> ~9 us)
// Tuning::scan_self_max (4096 tiles = 16.7 M elements): beyond it the scan of sums is its own launch

void launch_scan_u32_pair(const uint32_t* in0, uint64_t* out0, const uint32_t* in1, uint64_t* out1, size_t n,
                          void* temp, const Tuning& tn, hipStream_t st, uint64_t* host_total, const ScanExtra* x0,
                          const ScanExtra* x1) {
  if (n <= SCAN_SMALL_MAX) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, in0, out0, in1, out1, n, host_total);
  } else {
    launch_scan_u32(in0, n, out0, temp, tn, st, host_total, x0);
    if (in1) launch_scan_u32(in1, n, out1, temp, tn, st, nullptr, x1);
  }
}

bool scan_writes_ebase(size_t n) { return n > SCAN_SMALL_MAX; }

size_t scan_temp_bytes(size_t n) { return ((n + SCAN_TILE - 1) / SCAN_TILE + 4) * sizeof(uint64_t); }

void launch_scan_u32(const uint32_t* in, size_t n, uint64_t* out, void* temp, const Tuning& tn, hipStream_t st,
                     uint64_t* host_total, const ScanExtra* x) {
  const ScanEbase eb{x ? x->deg : nullptr, x ? x->degp : nullptr, x ? x->ebase : nullptr};
  const uint64_t* range = x ? x->range : nullptr;
  if (n <= SCAN_SMALL_MAX) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, st, in, out, (const uint32_t*)nullptr,
                       (uint64_t*)nullptr, n, host_total);
    return;
  }
  uint64_t* bsum = static_cast<uint64_t*>(temp);
  const size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) return;  // n == 0 is handled by the small path above
  // single-pass form while the look-back stays shallow (r02: 116 tiles 10.6 us against 4.9 + 7.9; 781 tiles 16.4 against
  // 5.0 + 9.4 — every 64 tiles are one more dependent round of descriptor loads)
  if (x && x->lb.epoch && x->lb.desc && x->lb.ticket && tn.scan_self_max != 0 && nb <= (size_t)(range ? 512 : 256)) {  // (tiles outside a range publish a zero at once: a trimmed launch may be longer)
    hipLaunchKernelGGL(scan_lookback_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, out, x->lb, host_total,
                       range, eb);
    return;
  }
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, bsum, range);
  if (nb <= tn.scan_self_max) {  // (a test sets scan_self_max = 0 to force the three-kernel form)
    hipLaunchKernelGGL(scan_downsweep_kernel<true>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, bsum, out,
                       host_total, range, eb);
  } else {
    hipLaunchKernelGGL(scan_of_sums_kernel, dim3(1), dim3(1024), 0, st, bsum, nb, out + n, host_total);
    hipLaunchKernelGGL(scan_downsweep_kernel<false>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, in, n, bsum, out,
                       (uint64_t*)nullptr, range, eb);
  }
}

}  // namespace sc
