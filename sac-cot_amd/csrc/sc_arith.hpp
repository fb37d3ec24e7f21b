// sc_arith.hpp — canonical fp32 arithmetic of the SAC-COT hot path, device side (gfx950).
//
// The reference (/root/reference/README.md:1-2) defines no numerics; the operation order below is the
// build decision frozen in SURVEY.md §8(a) / DESIGN.md §3.  Rules: fp32 only; every fused multiply-add
// is an explicit __builtin_fmaf; the translation unit is compiled with -ffp-contract=off so nothing else
// fuses; sqrt and '/' are the correctly rounded forms (hipcc default,
// -fhip-fp32-correctly-rounded-divide-sqrt); no OCML transcendental is called (sc_expf is a polynomial).
// oracle/saccot_oracle.c restates the same order independently in plain C; tests compare bit-for-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sc {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// Correctly rounded sqrt and divide.  NOT __fsqrt_rn / __fdiv_rn: in ROCm 7.2's __clang_hip_math.h those map to
// __ocml_native_sqrt_f32 (approximate) unless OCML_BASIC_ROUNDED_OPERATIONS is defined.  __builtin_sqrtf and
// the '/' operator take the IEEE expansions (v_sqrt_f32 / v_rcp_f32 seed + fma fix-up + v_div_fixup) as long
// as -fhip-fp32-correctly-rounded-divide-sqrt stays on (the hipcc default; build.py passes it explicitly).
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }

// Correctly rounded sqrt with fewer instructions, for the O(N^2) stage: the same algorithm LLVM emits for sqrtf
// (v_sqrt_f32 is within 1 ulp; the exact fma residuals of its two neighbours pick the correctly rounded value) minus
// the 2^32 pre-scaling that only inputs below 2^-96 need.  Such inputs (two points closer than ~3.6e-15) take the
// full sqrtf through a wave-uniform branch, so the result is identical to sqrt_rn for every input, including 0
// (NaN residual compares false, result 0) and +inf.
__device__ __forceinline__ float sqrt_rn_fast(float x) {
  if (__builtin_expect(__ballot(x < 0x1p-96f && x > 0.0f) != 0, 0)) return __builtin_sqrtf(x);
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
  const float rd = fma_(-sd, s, x), ru = fma_(-su, s, x);
  float r = (rd <= 0.0f) ? sd : s;
  r = (ru > 0.0f) ? su : r;
  return r;
}

// exp(x), x <= 0 (clamped at -87): k = rint(x log2e) via the 1.5*2^23 trick, two-term ln2 reduction,
// degree-6 Horner, scale through the exponent field.
__device__ __forceinline__ float sc_expf(float x) {
  const float LOG2E = 0x1.715476p+0f, LN2_HI = 0x1.62e400p-1f, LN2_LO = 0x1.7f7d1cp-20f;
  const float MAGIC = 12582912.0f;
  x = fmaxf(x, -87.0f);
  float kf = fma_(x, LOG2E, MAGIC);
  kf = kf - MAGIC;
  float r = fma_(kf, -LN2_HI, x);
  r = fma_(kf, -LN2_LO, r);
  float p = 0x1.6c16c2p-10f;
  p = fma_(p, r, 0x1.111112p-7f);
  p = fma_(p, r, 0x1.555556p-5f);
  p = fma_(p, r, 0x1.555556p-3f);
  p = fma_(p, r, 0.5f);
  p = fma_(p, r, 1.0f);
  p = fma_(p, r, 1.0f);
  int k = (int)kf;
  return p * __uint_as_float((uint32_t)(k + 127) << 23);
}

// |a - b| for 3-vectors: dx,dy,dz then n2 = fma(dz,dz, fma(dy,dy, dx*dx)), correctly rounded sqrt.
__device__ __forceinline__ float dist3(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return sqrt_rn(fma_(dz, dz, fma_(dy, dy, dx * dx)));
}
__device__ __forceinline__ float dist3_fast(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return sqrt_rn_fast(fma_(dz, dz, fma_(dy, dy, dx * dx)));
}

// Stage A pair test.  Returns the weight (0 when not an edge); `edge` gets the decision.
__device__ __forceinline__ float pair_weight(float dp, float dq, float d_thr, float min_len,
                                             float neg_inv2sig2, bool& edge) {
  float d = fabsf(dp - dq);
  edge = (d <= d_thr) && (dp >= min_len) && (dq >= min_len);
  return edge ? sc_expf((d * d) * neg_inv2sig2) : 0.0f;
}

// Stage C2/C3 inlier test: e_c = t_c + fma(r_c2,pz, fma(r_c1,py, fma(r_c0,px, -q_c))).
// (t_c + x is bit-identical to fmaf(t_c, 1, x); the chain is the k-ordered form of a 16x16x4 f32 MFMA
// with C = -q, kept so an MFMA variant stays bit-compatible.)
__device__ __forceinline__ float resid2(const float* __restrict__ M, float px, float py, float pz, float qx,
                                        float qy, float qz) {
  float ex = M[9] + fma_(M[2], pz, fma_(M[1], py, fma_(M[0], px, -qx)));
  float ey = M[10] + fma_(M[5], pz, fma_(M[4], py, fma_(M[3], px, -qy)));
  float ez = M[11] + fma_(M[8], pz, fma_(M[7], py, fma_(M[6], px, -qz)));
  return fma_(ez, ez, fma_(ey, ey, ex * ex));
}

__device__ __forceinline__ float dot3(const float* a, const float* b) {
  return fma_(a[2], b[2], fma_(a[1], b[1], a[0] * b[0]));
}
__device__ __forceinline__ void cross3(const float* a, const float* b, float* c) {
  c[0] = fma_(a[1], b[2], -(a[2] * b[1]));
  c[1] = fma_(a[2], b[0], -(a[0] * b[2]));
  c[2] = fma_(a[0], b[1], -(a[1] * b[0]));
}

constexpr int JACOBI_SWEEPS = 6;

// Stage C1: rigid transform of one triangle.  P,Q: three points each (row m = point m).
// One-sided (Hestenes) Jacobi on the columns of H = sum_m (p_m - pc)(q_m - qc)^T, fixed sweep count, then
// the two dominant singular pairs + cross products give R = V U^T with det +1.
__device__ __forceinline__ void kabsch3(const float P[9], const float Q[9], float Rt[12]) {
  const float THIRD = 0x1.555556p-2f;
  float pc[3], qc[3], a[3][3], b[3][3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    pc[c] = ((P[c] + P[3 + c]) + P[6 + c]) * THIRD;
    qc[c] = ((Q[c] + Q[3 + c]) + Q[6 + c]) * THIRD;
  }
#pragma unroll
  for (int m = 0; m < 3; m++)
#pragma unroll
    for (int c = 0; c < 3; c++) { a[m][c] = P[3 * m + c] - pc[c]; b[m][c] = Q[3 * m + c] - qc[c]; }
  float B[3][3], V[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) B[c][r] = fma_(a[2][r], b[2][c], fma_(a[1][r], b[1][c], a[0][r] * b[0][c]));
#pragma unroll 1
  for (int sweep = 0; sweep < JACOBI_SWEEPS; sweep++) {
#pragma unroll
    for (int pr = 0; pr < 3; pr++) {
      const int ip = (pr == 2) ? 1 : 0, iq = (pr == 0) ? 1 : 2;
      float alpha = dot3(B[ip], B[ip]), beta = dot3(B[iq], B[iq]), gamma = dot3(B[ip], B[iq]);
      if (gamma != 0.0f) {
        float zeta = div_rn(beta - alpha, gamma + gamma);
        float den = fabsf(zeta) + sqrt_rn(fma_(zeta, zeta, 1.0f));
        float tt = div_rn(1.0f, den);
        if (zeta < 0.0f) tt = -tt;
        float cs = div_rn(1.0f, sqrt_rn(fma_(tt, tt, 1.0f)));
        float sn = cs * tt;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          float x = B[ip][r], y = B[iq][r];
          B[ip][r] = fma_(-sn, y, cs * x);
          B[iq][r] = fma_(sn, x, cs * y);
          x = V[ip][r]; y = V[iq][r];
          V[ip][r] = fma_(-sn, y, cs * x);
          V[iq][r] = fma_(sn, x, cs * y);
        }
      }
    }
  }
  float n0 = dot3(B[0], B[0]), n1 = dot3(B[1], B[1]), n2 = dot3(B[2], B[2]);
  // largest and second-largest squared column norm, ties to the lower index (same decision tree as the oracle)
  int i1 = 0; float m1 = n0;
  if (n1 > m1) { i1 = 1; m1 = n1; }
  if (n2 > m1) { i1 = 2; m1 = n2; }
  int i2 = (i1 == 0) ? 1 : 0;
  {
    const int c = 3 - i1 - i2;  // the remaining index
    float nc = (c == 0) ? n0 : (c == 1 ? n1 : n2);
    float ni2 = (i2 == 0) ? n0 : (i2 == 1 ? n1 : n2);
    // the oracle scans c = 0,1,2 skipping i1 and the initial i2; only one candidate remains
    if (nc > ni2) i2 = c;
  }
  float b1[3], b2[3], v1[3], v2[3];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    b1[r] = (i1 == 0) ? B[0][r] : (i1 == 1 ? B[1][r] : B[2][r]);
    b2[r] = (i2 == 0) ? B[0][r] : (i2 == 1 ? B[1][r] : B[2][r]);
    v1[r] = (i1 == 0) ? V[0][r] : (i1 == 1 ? V[1][r] : V[2][r]);
    v2[r] = (i2 == 0) ? V[0][r] : (i2 == 1 ? V[1][r] : V[2][r]);
  }
  float s1 = sqrt_rn(dot3(b1, b1)), s2 = sqrt_rn(dot3(b2, b2));
  float u1[3], u2[3], u3[3], v3[3];
#pragma unroll
  for (int r = 0; r < 3; r++) { u1[r] = div_rn(b1[r], s1); u2[r] = div_rn(b2[r], s2); }
  cross3(u1, u2, u3);
  cross3(v1, v2, v3);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) Rt[3 * r + c] = fma_(v3[r], u3[c], fma_(v2[r], u2[c], v1[r] * u1[c]));
#pragma unroll
  for (int r = 0; r < 3; r++)
    Rt[9 + r] = qc[r] - fma_(Rt[3 * r + 2], pc[2], fma_(Rt[3 * r + 1], pc[1], Rt[3 * r] * pc[0]));
}

__device__ __forceinline__ bool finite12(const float* M) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < 12; c++) ok = ok && (fabsf(M[c]) < __builtin_inff());  // false for inf and NaN
  return ok;
}

}  // namespace sc
