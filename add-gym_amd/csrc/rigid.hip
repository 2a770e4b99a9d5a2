// Rigid-body step of the articulated robot behind the engine plugin API (BaseScene.step + BaseEntity.control_dofs_position,
// add_gym/engine/base_engine.py:93-455): one launch per CONTROL step, `substeps` physics steps inside.
//
// The reference delegates this to Genesis / MuJoCo-Warp (engine/genesis_engine.py, engine/mjwarp_engine.py:807-851,
// 1554-1611); the algorithm here is this repo's own (oracle/rigid.py is its float64 restatement, parity is pinned to that
// and to physical invariants, not to the reference: DESIGN.md).
//
// Per substep h:  articulated-body algorithm (Featherstone) over the kinematic tree with a floating base, in body
// coordinates, with every stiff term folded into the inertia it acts on so that ONE O(bodies) sweep per substep is stable at
// millisecond steps and no constraint solver iterates:
//   * joint PD + damping + armature:  D_i += armature + h (kv + damping) + h^2 kp   (stable PD; plain clamp when saturated)
//   * ground contacts of the body's collision spheres (spring-damper normal, regularised Coulomb friction), linearly
//     implicit:  IA_i += J^T (h^2 K + h C) J,   pA_i -= J^T (f0 - h K v_p)
// then semi-implicit Euler.
//
// Mapping to gfx950: one LANE per environment, one 64-lane workgroup per CU.  The tree is walked in depth-first order, so
// the inward pass only ever holds the current chain's articulated inertia in registers plus one accumulator per BRANCH body
// (pelvis, torso) in LDS.  Every lane of a wave visits the same body at the same time: body constants and the traversal
// tables are wave-uniform scalar loads, LDS arrays are [field][lane] (conflict-free), the state rows are staged through LDS
// with coalesced loads/stores.  Per-lane LDS: pose 36 + velocity 36 + target 32 + sin/cos 64 + U 192 + 1/D, u 64 +
// 4 branch accumulators x 27 floats = 532 floats (133 KB per workgroup) + 11 KB of model constants staged once per workgroup;
// body velocities / up-vectors live in private (scratch) arrays that stay in L2.
#include <mutex>
#include "common.h"
#include "record.h"
#include "philox.h"

namespace {

constexpr int MAXB = 32;       // bodies
constexpr int WG = 64;         // environments (lanes) per workgroup
constexpr int BW = ADDHIP_RIGID_BODY_W;
constexpr int TW = ADDHIP_RIGID_TOPO_W;
constexpr int MAX_SLOTS = 4;   // branch bodies (more than one child)
constexpr int MAXP = 384;      // collision points

typedef const float __attribute__((address_space(4))) cfloat;
typedef const int __attribute__((address_space(4))) cint;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// general 3x3, row major
struct M3 { float m[9]; };
// symmetric 3x3: xx xy xz yy yz zz
struct S3 { float xx, xy, xz, yy, yz, zz; };

__device__ __forceinline__ V3 mul(const M3& a, V3 v) {
  return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
__device__ __forceinline__ V3 mulT(const M3& a, V3 v) {
  return {a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z, a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
__device__ __forceinline__ V3 mul(const S3& a, V3 v) {
  return {a.xx * v.x + a.xy * v.y + a.xz * v.z, a.xy * v.x + a.yy * v.y + a.yz * v.z, a.xz * v.x + a.yz * v.y + a.zz * v.z};
}
__device__ __forceinline__ M3 matmul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}
__device__ __forceinline__ M3 transpose(const M3& a) { return {{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}}; }
__device__ __forceinline__ M3 full(const S3& s) { return {{s.xx, s.xy, s.xz, s.xy, s.yy, s.yz, s.xz, s.yz, s.zz}}; }
// skew(r) * M   (rows: r x (column-wise))
__device__ __forceinline__ M3 skew_mul(V3 r, const M3& a) {
  M3 o;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const V3 c = cross(r, V3{a.m[j], a.m[3 + j], a.m[6 + j]});
    o.m[j] = c.x; o.m[3 + j] = c.y; o.m[6 + j] = c.z;
  }
  return o;
}
// M * skew(r)   (row i of the result = row_i(M) x r ... as  (row x r)?  row * skew(r) = -(r x row)^T = (row x r)
__device__ __forceinline__ M3 mul_skew(const M3& a, V3 r) {
  M3 o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const V3 c = cross(V3{a.m[3 * i], a.m[3 * i + 1], a.m[3 * i + 2]}, r);
    // (row * skew(r))_j = sum_k row_k skew(r)_{kj};  skew(r) v = r x v  =>  row * skew(r) = (skew(r)^T row)^T = -(r x row) = row x r
    o.m[3 * i] = c.x; o.m[3 * i + 1] = c.y; o.m[3 * i + 2] = c.z;
  }
  return o;
}
// R S R^T for symmetric S (result symmetric)
__device__ __forceinline__ S3 rot_sym(const M3& R, const S3& s) {
  const M3 t = matmul(R, full(s));
  S3 o;
  o.xx = t.m[0] * R.m[0] + t.m[1] * R.m[1] + t.m[2] * R.m[2];
  o.xy = t.m[0] * R.m[3] + t.m[1] * R.m[4] + t.m[2] * R.m[5];
  o.xz = t.m[0] * R.m[6] + t.m[1] * R.m[7] + t.m[2] * R.m[8];
  o.yy = t.m[3] * R.m[3] + t.m[4] * R.m[4] + t.m[5] * R.m[5];
  o.yz = t.m[3] * R.m[6] + t.m[4] * R.m[7] + t.m[5] * R.m[8];
  o.zz = t.m[6] * R.m[6] + t.m[7] * R.m[7] + t.m[8] * R.m[8];
  return o;
}

// articulated inertia [[A, B], [B^T, C]] (A, C symmetric) and bias force [n; f]
struct ArtI { S3 A; M3 B; S3 C; };
struct Sp6 { V3 a, l; };  // angular / linear part of a spatial vector

__device__ __forceinline__ float comp(V3 v, int ax) { return ax == 0 ? v.x : (ax == 1 ? v.y : v.z); }
__device__ __forceinline__ V3 unit(int ax) { return {ax == 0 ? 1.f : 0.f, ax == 1 ? 1.f : 0.f, ax == 2 ? 1.f : 0.f}; }

// child -> parent rotation  R = Rfix * Rot(axis, theta)
template <typename P>
__device__ __forceinline__ M3 joint_rot(P rf, int ax, float s, float c) {
  M3 R;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a = rf[3 * i], b = rf[3 * i + 1], d = rf[3 * i + 2];
    if (ax == 0) { R.m[3 * i] = a; R.m[3 * i + 1] = c * b + s * d; R.m[3 * i + 2] = c * d - s * b; }
    else if (ax == 1) { R.m[3 * i] = c * a - s * d; R.m[3 * i + 1] = b; R.m[3 * i + 2] = s * a + c * d; }
    else { R.m[3 * i] = c * a + s * b; R.m[3 * i + 1] = c * b - s * a; R.m[3 * i + 2] = d; }
  }
  return R;
}

// LDS field offsets (floats per lane)
constexpr int F_POSE = 0, F_VEL = 36, F_TGT = 72, F_SIN = 104, F_COS = 136, F_U = 168, F_DINV = 360, F_UU = 392, F_SLOT = 424;
constexpr int F_TOTAL = F_SLOT + MAX_SLOTS * 27;
constexpr int LDS_FLOATS = F_TOTAL * WG;

__device__ __forceinline__ void store_art(float* p, const ArtI& I, const Sp6& f) {  // p points at lds[...][lane], stride WG
  const float v[27] = {I.A.xx, I.A.xy, I.A.xz, I.A.yy, I.A.yz, I.A.zz, I.B.m[0], I.B.m[1], I.B.m[2], I.B.m[3], I.B.m[4], I.B.m[5], I.B.m[6], I.B.m[7], I.B.m[8],
                       I.C.xx, I.C.xy, I.C.xz, I.C.yy, I.C.yz, I.C.zz, f.a.x, f.a.y, f.a.z, f.l.x, f.l.y, f.l.z};
#pragma unroll
  for (int i = 0; i < 27; ++i) p[i * WG] = v[i];
}
__device__ __forceinline__ void add_art(const float* p, ArtI& I, Sp6& f) {
  I.A.xx += p[0]; I.A.xy += p[WG]; I.A.xz += p[2 * WG]; I.A.yy += p[3 * WG]; I.A.yz += p[4 * WG]; I.A.zz += p[5 * WG];
#pragma unroll
  for (int i = 0; i < 9; ++i) I.B.m[i] += p[(6 + i) * WG];
  I.C.xx += p[15 * WG]; I.C.xy += p[16 * WG]; I.C.xz += p[17 * WG]; I.C.yy += p[18 * WG]; I.C.yz += p[19 * WG]; I.C.zz += p[20 * WG];
  f.a.x += p[21 * WG]; f.a.y += p[22 * WG]; f.a.z += p[23 * WG]; f.l.x += p[24 * WG]; f.l.y += p[25 * WG]; f.l.z += p[26 * WG];
}

__global__ __launch_bounds__(WG) void rigid_step_kernel(addhip_rigid_model_t M, float* __restrict__ sim_pose, float* __restrict__ sim_vel,
                                                        const float* __restrict__ target, int tstride, int n, unsigned char* __restrict__ contact_flag,
                                                        unsigned* __restrict__ contact_bits) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int env0 = blockIdx.x * WG;
  const int env = env0 + lane;
  const int live = min(WG, n - env0);
#define LD(f, i) lds[((f) + (i)) * WG + lane]

  // ---- stage the state rows through LDS: coalesced global access, [field][lane] in LDS
  for (int idx = lane; idx < live * 36; idx += WG) {
    const int row = idx / 36, col = idx - row * 36;
    lds[(F_POSE + col) * WG + row] = sim_pose[(size_t)env0 * 36 + idx];
    lds[(F_VEL + col) * WG + row] = sim_vel[(size_t)env0 * 36 + idx];
  }
  for (int idx = lane; idx < live * 32; idx += WG) {
    const int row = idx >> 5, col = idx & 31;
    lds[(F_TGT + col) * WG + row] = col < 29 ? target[(size_t)(env0 + row) * tstride + col] : 0.f;
  }
  __syncthreads();
  // the model tables are read-only for the whole launch and every lane of a wave reads the same element: address them through
  // the constant address space, so that they arrive by scalar loads (SGPR operands, scalar cache) instead of per-lane loads
  const cfloat* const cbody = (cfloat*)(uintptr_t)M.body;
  const cint* const ctopo = (cint*)(uintptr_t)M.topo;
  const cfloat* const cpts = (cfloat*)(uintptr_t)M.points;
  const bool on = lane < live;  // lanes past the last env run on zeros (kept in step: no divergence, never written back)
  if (!on) {
    for (int c = 0; c < 36; ++c) { LD(F_POSE, c) = (c == 3) ? 1.f : 0.f; LD(F_VEL, c) = 0.f; }
    LD(F_POSE, 2) = 10.f;
    for (int c = 0; c < 32; ++c) LD(F_TGT, c) = 0.f;
  }

  const int nb = M.num_bodies;
  const float h = M.dt / (float)M.substeps;
  const float kc = M.contact_stiffness, cn = M.contact_damping, veps = M.friction_vel_eps;
  // domain randomisation: per-env PD gain multiplier and ground friction (both 1 x / the model's value without it)
  const float gscale = (M.env_scale && on) ? M.env_scale[2 * env] : 1.f;
  const float mu = (M.env_scale && on) ? M.env_scale[2 * env + 1] : M.friction;
  // private per-body arrays (scratch): spatial velocity, then (pass 3) spatial acceleration; world up-vector in body coords; height
  float bv[MAXB][6], bnz[MAXB][3], bh[MAXB];
  unsigned touch = 0;

  V3 pos{LD(F_POSE, 0), LD(F_POSE, 1), LD(F_POSE, 2)};
  float qw = LD(F_POSE, 3), qx = LD(F_POSE, 4), qy = LD(F_POSE, 5), qz = LD(F_POSE, 6);
  V3 vw{LD(F_VEL, 0), LD(F_VEL, 1), LD(F_VEL, 2)}, ww{LD(F_VEL, 3), LD(F_VEL, 4), LD(F_VEL, 5)};

  for (int sub = 0; sub < M.substeps; ++sub) {
    touch = 0;
    // root rotation (body -> world)
    M3 R0 = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
              2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
              2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)}};
    // ---------------- pass 1 (outward): velocities, up-vector, height.  In depth-first order the parent of body k is k-1 along
    // a chain: its state is carried in registers; only the first body of a limb re-reads its (branch) parent from scratch.
    V3 cw, cv, cnz; float ch;  // state of body k-1
    {
      cw = mulT(R0, ww); cv = mulT(R0, vw);
      cnz = {R0.m[6], R0.m[7], R0.m[8]}; ch = pos.z;
      bv[0][0] = cw.x; bv[0][1] = cw.y; bv[0][2] = cw.z; bv[0][3] = cv.x; bv[0][4] = cv.y; bv[0][5] = cv.z;
      bnz[0][0] = cnz.x; bnz[0][1] = cnz.y; bnz[0][2] = cnz.z;
      bh[0] = ch;
    }
    for (int k = 1; k < nb; ++k) {
      const cfloat* bc = cbody + k * BW;
      const cint* tp = ctopo + k * TW;
      const int par = tp[0], ax = tp[1], dof = tp[2];
      const float q = LD(F_POSE, 7 + dof), qd = LD(F_VEL, 6 + dof);
      float s, c;
      if (sub == 0) {  // later substeps: pass 3 of the previous substep advanced sin/cos by the angle increment
        sincosf(q, &s, &c);
        LD(F_SIN, k) = s; LD(F_COS, k) = c;
      } else {
        s = LD(F_SIN, k); c = LD(F_COS, k);
      }
      const M3 R = joint_rot(bc + 3, ax, s, c);
      const V3 r{bc[0], bc[1], bc[2]};
      if (par != k - 1) {  // wave-uniform
        cw = {bv[par][0], bv[par][1], bv[par][2]}; cv = {bv[par][3], bv[par][4], bv[par][5]};
        cnz = {bnz[par][0], bnz[par][1], bnz[par][2]}; ch = bh[par];
      }
      V3 w = mulT(R, cw);
      const V3 vl = mulT(R, cv - cross(r, cw));
      if (ax == 0) w.x += qd; else if (ax == 1) w.y += qd; else w.z += qd;
      bv[k][0] = w.x; bv[k][1] = w.y; bv[k][2] = w.z; bv[k][3] = vl.x; bv[k][4] = vl.y; bv[k][5] = vl.z;
      const V3 nz = mulT(R, cnz);
      bnz[k][0] = nz.x; bnz[k][1] = nz.y; bnz[k][2] = nz.z;
      ch = ch + dot(cnz, r);
      bh[k] = ch;
      cw = w; cv = vl; cnz = nz;
    }
    // ---------------- pass 2 (inward, reverse depth-first order)
    ArtI carryI; Sp6 carryP;  // contribution of body k+1 to its chain parent k, in k's coordinates
    for (int s = 0; s < MAX_SLOTS * 27; ++s) LD(F_SLOT, s) = 0.f;
    // the pass-1 results of the NEXT body are fetched from scratch while the current one is worked on
    float nx[10];
#pragma unroll
    for (int i = 0; i < 6; ++i) nx[i] = bv[nb - 1][i];
    nx[6] = bnz[nb - 1][0]; nx[7] = bnz[nb - 1][1]; nx[8] = bnz[nb - 1][2]; nx[9] = bh[nb - 1];
    for (int k = nb - 1; k >= 0; --k) {
      const cfloat* bc = cbody + k * BW;
      const cint* tp = ctopo + k * TW;
      const int par = tp[0], ax = tp[1], dof = tp[2], nchild = tp[3], slot = tp[4], pt0 = tp[5], npt = tp[6], link = tp[7];
      const float mass = bc[12];
      const V3 mc{bc[13], bc[14], bc[15]};
      const V3 w{nx[0], nx[1], nx[2]}, vl{nx[3], nx[4], nx[5]};
      const V3 nz{nx[6], nx[7], nx[8]};
      const float hk = nx[9];
      if (k > 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) nx[i] = bv[k - 1][i];
        nx[6] = bnz[k - 1][0]; nx[7] = bnz[k - 1][1]; nx[8] = bnz[k - 1][2]; nx[9] = bh[k - 1];
      }
      // rigid-body inertia and bias:  pA = v x* (I v) - I a_g
      ArtI I;
      I.A = {bc[16], bc[17], bc[18], bc[19], bc[20], bc[21]};
      I.B = {{0.f, -mc.z, mc.y, mc.z, 0.f, -mc.x, -mc.y, mc.x, 0.f}};
      I.C = {mass, 0.f, 0.f, mass, 0.f, mass};
      Sp6 p;
      {
        const V3 hn = mul(I.A, w) + cross(mc, vl);        // angular momentum about the body origin
        const V3 hl = mass * vl - cross(mc, w);           // linear momentum
        const V3 gl = (-M.gravity) * nz;
        p.a = cross(w, hn) + cross(vl, hl) - cross(mc, gl);
        p.l = cross(w, hl) - mass * gl;
      }
      if (nchild == 1 && slot < 0) {
        I.A.xx += carryI.A.xx; I.A.xy += carryI.A.xy; I.A.xz += carryI.A.xz; I.A.yy += carryI.A.yy; I.A.yz += carryI.A.yz; I.A.zz += carryI.A.zz;
#pragma unroll
        for (int i = 0; i < 9; ++i) I.B.m[i] += carryI.B.m[i];
        I.C.xx += carryI.C.xx; I.C.xy += carryI.C.xy; I.C.xz += carryI.C.xz; I.C.yy += carryI.C.yy; I.C.yz += carryI.C.yz; I.C.zz += carryI.C.zz;
        p.a = p.a + carryP.a; p.l = p.l + carryP.l;
      } else if (slot >= 0) {
        add_art(&LD(F_SLOT, slot * 27), I, p);
      }
      // ground contacts of this body's collision spheres (skipped for the whole wave while the body's bounding sphere is clear
      // of the ground in every lane)
      const bool near = hk < bc[29];
      if (__any(near))
      for (int j = 0; j < npt; ++j) {
        const cfloat* pt = cpts + (pt0 + j) * 4;
        const V3 r{pt[0], pt[1], pt[2]};
        const float d = pt[3] - (hk + dot(nz, r));
        if (d > 0.f) {
          touch |= 1u << link;
          const V3 vp = vl + cross(w, r);
          const float vn = dot(nz, vp);
          const V3 vt = vp - vn * nz;
          const float fn0 = fmaxf(kc * d - cn * vn, 0.f);
          const float ct = mu * fn0 / fmaxf(sqrtf(dot(vt, vt)), veps);
          const float fne = fmaxf(fn0 - h * kc * vn, 0.f);
          const V3 f = fne * nz - ct * vt;
          const float an = h * h * kc + h * cn - h * ct, at = h * ct;   // A = an nz nz^T + at 1
          const S3 Am{an * nz.x * nz.x + at, an * nz.x * nz.y, an * nz.x * nz.z, an * nz.y * nz.y + at, an * nz.y * nz.z, an * nz.z * nz.z + at};
          const M3 rA = skew_mul(r, full(Am));          // B += r x A
          const M3 rArT = mul_skew(rA, V3{-r.x, -r.y, -r.z});  // (r x A) skew(r)^T
#pragma unroll
          for (int i = 0; i < 9; ++i) I.B.m[i] += rA.m[i];
          I.A.xx += rArT.m[0]; I.A.xy += rArT.m[1]; I.A.xz += rArT.m[2]; I.A.yy += rArT.m[4]; I.A.yz += rArT.m[5]; I.A.zz += rArT.m[8];
          I.C.xx += Am.xx; I.C.xy += Am.xy; I.C.xz += Am.xz; I.C.yy += Am.yy; I.C.yz += Am.yz; I.C.zz += Am.zz;
          p.a = p.a - cross(r, f);
          p.l = p.l - f;
        }
      }
      if (k == 0) { carryI = I; carryP = p; break; }
      // joint: PD torque with the implicit diagonal (stable PD unless the torque clamp is active), limit spring
      const float q = LD(F_POSE, 7 + dof), qd = LD(F_VEL, 6 + dof);
      const float lo = bc[22], hi = bc[23], damp = bc[24], arm = bc[25], flim = fminf(bc[26], M.max_torque), kp = gscale * bc[27], kv = gscale * bc[28];
      const float tgt = fminf(fmaxf(LD(F_TGT, dof), lo + M.limit_margin), hi - M.limit_margin);
      const float tpd = kp * (tgt - q) - kv * qd;
      float tau, dadd;
      if (fabsf(tpd) > flim) { tau = fminf(fmaxf(tpd, -flim), flim) - damp * qd; dadd = arm + h * damp; }
      else { tau = kp * (tgt - q - h * qd) - (kv + damp) * qd; dadd = arm + h * (kv + damp) + h * h * kp; }
      if (q < lo) { tau += M.limit_stiffness * (lo - q - h * qd); dadd += h * h * M.limit_stiffness; }
      if (q > hi) { tau += M.limit_stiffness * (hi - q - h * qd); dadd += h * h * M.limit_stiffness; }
      // U = IA S (S = angular unit axis), D, u
      V3 Ua, Ul;
      if (ax == 0) { Ua = {I.A.xx, I.A.xy, I.A.xz}; Ul = {I.B.m[0], I.B.m[1], I.B.m[2]}; }
      else if (ax == 1) { Ua = {I.A.xy, I.A.yy, I.A.yz}; Ul = {I.B.m[3], I.B.m[4], I.B.m[5]}; }
      else { Ua = {I.A.xz, I.A.yz, I.A.zz}; Ul = {I.B.m[6], I.B.m[7], I.B.m[8]}; }
      const float Dinv = 1.f / (comp(Ua, ax) + dadd);
      const float u = tau - comp(p.a, ax);
      LD(F_U, 6 * k) = Ua.x; LD(F_U, 6 * k + 1) = Ua.y; LD(F_U, 6 * k + 2) = Ua.z; LD(F_U, 6 * k + 3) = Ul.x; LD(F_U, 6 * k + 4) = Ul.y; LD(F_U, 6 * k + 5) = Ul.z;
      LD(F_DINV, k) = Dinv; LD(F_UU, k) = u;
      // Ia = IA - U U^T / D
      I.A.xx -= Dinv * Ua.x * Ua.x; I.A.xy -= Dinv * Ua.x * Ua.y; I.A.xz -= Dinv * Ua.x * Ua.z; I.A.yy -= Dinv * Ua.y * Ua.y; I.A.yz -= Dinv * Ua.y * Ua.z; I.A.zz -= Dinv * Ua.z * Ua.z;
      I.C.xx -= Dinv * Ul.x * Ul.x; I.C.xy -= Dinv * Ul.x * Ul.y; I.C.xz -= Dinv * Ul.x * Ul.z; I.C.yy -= Dinv * Ul.y * Ul.y; I.C.yz -= Dinv * Ul.y * Ul.z; I.C.zz -= Dinv * Ul.z * Ul.z;
      {
        const float ua[3] = {Ua.x, Ua.y, Ua.z}, ul[3] = {Ul.x, Ul.y, Ul.z};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) I.B.m[3 * i + j] -= Dinv * ua[i] * ul[j];
      }
      // pa = pA + Ia c + U u / D,   c = [w x e; vl x e] qd
      const V3 e = unit(ax);
      const V3 ca = qd * cross(w, e), cl = qd * cross(vl, e);
      const float ud = u * Dinv;
      p.a = p.a + mul(I.A, ca) + mul(I.B, cl) + ud * Ua;
      p.l = p.l + mulT(I.B, ca) + mul(I.C, cl) + ud * Ul;
      // to the parent's coordinates:  X^T Ia X,  X^T pa   (X = rot(R^T) xlt(r))
      const M3 R = joint_rot(bc + 3, ax, LD(F_SIN, k), LD(F_COS, k));
      const V3 r{bc[0], bc[1], bc[2]};
      ArtI P;
      P.A = rot_sym(R, I.A);
      P.C = rot_sym(R, I.C);
      const M3 Bp = matmul(matmul(R, I.B), transpose(R));
      const M3 rC = skew_mul(r, full(P.C));
      M3 Bn;
#pragma unroll
      for (int i = 0; i < 9; ++i) Bn.m[i] = Bp.m[i] + rC.m[i];             // B_p = B' + r x C'
      const M3 t1 = skew_mul(r, transpose(Bn));                             // r x B_p^T
      const M3 t2 = mul_skew(Bp, r);                                        // B' skew(r)
      P.A.xx += t1.m[0] - t2.m[0]; P.A.xy += t1.m[1] - t2.m[1]; P.A.xz += t1.m[2] - t2.m[2];
      P.A.yy += t1.m[4] - t2.m[4]; P.A.yz += t1.m[5] - t2.m[5]; P.A.zz += t1.m[8] - t2.m[8];
      P.B = Bn;
      Sp6 pp;
      pp.l = mul(R, p.l);
      pp.a = mul(R, p.a) + cross(r, pp.l);
      const int pslot = ctopo[par * TW + 4];
      if (pslot >= 0) {
        float* sp = &LD(F_SLOT, pslot * 27);
        ArtI Z; Sp6 z0;
        Z.A = {0, 0, 0, 0, 0, 0}; Z.C = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 9; ++i) Z.B.m[i] = 0.f;
        z0.a = {0, 0, 0}; z0.l = {0, 0, 0};
        add_art(sp, Z, z0);
        Z.A.xx += P.A.xx; Z.A.xy += P.A.xy; Z.A.xz += P.A.xz; Z.A.yy += P.A.yy; Z.A.yz += P.A.yz; Z.A.zz += P.A.zz;
#pragma unroll
        for (int i = 0; i < 9; ++i) Z.B.m[i] += P.B.m[i];
        Z.C.xx += P.C.xx; Z.C.xy += P.C.xy; Z.C.xz += P.C.xz; Z.C.yy += P.C.yy; Z.C.yz += P.C.yz; Z.C.zz += P.C.zz;
        z0.a = z0.a + pp.a; z0.l = z0.l + pp.l;
        store_art(sp, Z, z0);
      } else {
        carryI = P; carryP = pp;
      }
    }
    // ---------------- floating base: a0 = -IA0^-1 pA0 (6x6 SPD, Cholesky)
    float a0[6];
    {
      const ArtI& I = carryI;
      float A[6][6];
      A[0][0] = I.A.xx; A[0][1] = I.A.xy; A[0][2] = I.A.xz; A[1][1] = I.A.yy; A[1][2] = I.A.yz; A[2][2] = I.A.zz;
      A[3][3] = I.C.xx; A[3][4] = I.C.xy; A[3][5] = I.C.xz; A[4][4] = I.C.yy; A[4][5] = I.C.yz; A[5][5] = I.C.zz;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i][3 + j] = I.B.m[3 * i + j];
      float b[6] = {-carryP.a.x, -carryP.a.y, -carryP.a.z, -carryP.l.x, -carryP.l.y, -carryP.l.z};
      // upper-triangular Cholesky A = L L^T stored in the upper part as L^T
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = i; j < 6; ++j) {
          float sum = A[i][j];
#pragma unroll
          for (int t = 0; t < i; ++t) sum -= A[t][i] * A[t][j];
          A[i][j] = (j == i) ? sqrtf(fmaxf(sum, 1e-20f)) : sum / A[i][i];
        }
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {  // L y = b
        float sum = b[i];
#pragma unroll
        for (int t = 0; t < i; ++t) sum -= A[t][i] * b[t];
        b[i] = sum / A[i][i];
      }
#pragma unroll
      for (int i = 5; i >= 0; --i) {  // L^T x = y
        float sum = b[i];
#pragma unroll
        for (int t = i + 1; t < 6; ++t) sum -= A[i][t] * b[t];
        b[i] = sum / A[i][i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) a0[i] = b[i];
    }
    // ---------------- pass 3 (outward): accelerations; joints are integrated as they are visited
    // the parent's acceleration is carried in registers along a chain; branch bodies also park theirs in bv[] (their own
    // velocity is no longer needed by then) for the limbs that start from them
    float root_a[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { root_a[i] = a0[i]; bv[0][i] = a0[i]; }
    V3 caa{a0[0], a0[1], a0[2]}, cal{a0[3], a0[4], a0[5]};
    for (int k = 1; k < nb; ++k) {
      const cfloat* bc = cbody + k * BW;
      const cint* tp = ctopo + k * TW;
      const int par = tp[0], ax = tp[1], dof = tp[2], slot = tp[4];
      const M3 R = joint_rot(bc + 3, ax, LD(F_SIN, k), LD(F_COS, k));
      const V3 r{bc[0], bc[1], bc[2]};
      const V3 w{bv[k][0], bv[k][1], bv[k][2]}, vl{bv[k][3], bv[k][4], bv[k][5]};
      if (par != k - 1) { caa = {bv[par][0], bv[par][1], bv[par][2]}; cal = {bv[par][3], bv[par][4], bv[par][5]}; }
      const float qd = LD(F_VEL, 6 + dof);
      const V3 e = unit(ax);
      V3 aa = mulT(R, caa) + qd * cross(w, e);
      const V3 al = mulT(R, cal - cross(r, caa)) + qd * cross(vl, e);
      const V3 Ua{LD(F_U, 6 * k), LD(F_U, 6 * k + 1), LD(F_U, 6 * k + 2)}, Ul{LD(F_U, 6 * k + 3), LD(F_U, 6 * k + 4), LD(F_U, 6 * k + 5)};
      const float qdd = (LD(F_UU, k) - dot(Ua, aa) - dot(Ul, al)) * LD(F_DINV, k);
      if (ax == 0) aa.x += qdd; else if (ax == 1) aa.y += qdd; else aa.z += qdd;
      if (slot >= 0) { bv[k][0] = aa.x; bv[k][1] = aa.y; bv[k][2] = aa.z; bv[k][3] = al.x; bv[k][4] = al.y; bv[k][5] = al.z; }
      caa = aa; cal = al;
      const float qdn = qd + h * qdd;
      LD(F_VEL, 6 + dof) = qdn;
      const float dq = h * qdn;
      LD(F_POSE, 7 + dof) = LD(F_POSE, 7 + dof) + dq;
      {  // sin/cos of the new angle by the addition theorem (|dq| = h |qd| << 1: 7th-order series, error < 1e-9 for |dq| < 0.2)
        const float d2 = dq * dq;
        const float sd = dq * (1.f + d2 * (-1.f / 6.f + d2 * (1.f / 120.f - d2 * (1.f / 5040.f))));
        const float cd = 1.f + d2 * (-0.5f + d2 * (1.f / 24.f - d2 * (1.f / 720.f)));
        const float s0 = LD(F_SIN, k), c0 = LD(F_COS, k);
        LD(F_SIN, k) = s0 * cd + c0 * sd;
        LD(F_COS, k) = c0 * cd - s0 * sd;
      }
    }
    // ---------------- root: semi-implicit Euler (classical linear acceleration = R a_lin + w x v)
    {
      const V3 aw = mul(R0, V3{root_a[0], root_a[1], root_a[2]});
      const V3 al = mul(R0, V3{root_a[3], root_a[4], root_a[5]}) + cross(ww, vw);
      ww = ww + h * aw;
      vw = vw + h * al;
      pos = pos + h * vw;
      const float wn = sqrtf(dot(ww, ww));
      const float half = 0.5f * h * wn;
      const float kq = wn > 1e-12f ? sinf(half) / wn : 0.5f * h;
      const float dw = cosf(half), dx = kq * ww.x, dy = kq * ww.y, dz = kq * ww.z;
      const float nw = dw * qw - dx * qx - dy * qy - dz * qz;
      const float nx = dw * qx + dx * qw + dy * qz - dz * qy;
      const float ny = dw * qy - dx * qz + dy * qw + dz * qx;
      const float nzq = dw * qz + dx * qy - dy * qx + dz * qw;
      const float inv = rsqrtf(nw * nw + nx * nx + ny * ny + nzq * nzq);
      qw = nw * inv; qx = nx * inv; qy = ny * inv; qz = nzq * inv;
    }
  }
  // ---- write back
  LD(F_POSE, 0) = pos.x; LD(F_POSE, 1) = pos.y; LD(F_POSE, 2) = pos.z;
  LD(F_POSE, 3) = qw; LD(F_POSE, 4) = qx; LD(F_POSE, 5) = qy; LD(F_POSE, 6) = qz;
  LD(F_VEL, 0) = vw.x; LD(F_VEL, 1) = vw.y; LD(F_VEL, 2) = vw.z; LD(F_VEL, 3) = ww.x; LD(F_VEL, 4) = ww.y; LD(F_VEL, 5) = ww.z;
  LD(F_VEL, 35) = 0.f;
  if (on) {
    if (contact_bits) contact_bits[env] = touch;
    if (contact_flag) contact_flag[env] = (touch & M.termination_mask) ? 1 : 0;
  }
  __syncthreads();
  for (int idx = lane; idx < live * 36; idx += WG) {
    const int row = idx / 36, col = idx - row * 36;
    sim_pose[(size_t)env0 * 36 + idx] = lds[(F_POSE + col) * WG + row];
    sim_vel[(size_t)env0 * 36 + idx] = lds[(F_VEL + col) * WG + row];
  }
#undef LD
}


// ---------------------------------------------------------------------------------------------------------------------
// Four lanes per environment ("limb lanes").  The one-lane kernel above is bound by the instruction stream of a single wave
// (30 bodies x 3 passes x 4 substeps, one wave per CU and only num_envs / 64 CUs busy).  Here the depth-first body order is cut
// into its chains (maximal runs whose parent is the previous body: for G1 left leg | right leg | waist + left arm | right arm)
// and each chain gets its own lane of a quad, so that a wave walks 10 body-steps instead of 30 and four times as many waves
// share the work.  Step s of a pass is local body s - start of every chain (a chain hanging off another chain's body starts
// one step after that body).  What crosses lanes, by quad-wide shuffles at the few steps where chains meet:
//   pass 1 / pass 3 (outward): the attach body's carried state (velocity, up-vector, height / acceleration);
//   pass 2 (inward): a chain's articulated inertia and bias force, expressed in its attach body's frame, is added to that
//   body's before it is processed; the chains hanging off the root are summed over the quad, and all four lanes then do the
//   root's own work and the 6x6 solve redundantly (no divergence, no broadcast).
// The model tables cannot be scalar loads any more (the lanes of a wave are on four different bodies): they are staged in
// LDS once per workgroup (body rows padded to 33 floats so that the four rows fall on different banks).
// Chain table (addhip_rigid_model_t.chains, 4 x 16 int32): len, start step, attach lane (-1: root), body indices.
constexpr int L4_S = 10;                 // steps per pass = max(start + len) over the chains
constexpr int L4_ENVS = 16;              // environments per 64-lane workgroup
constexpr int L4_BW = BW + 1;
constexpr int L4_STATE = L4_ENVS * (36 + 36 + 32);
constexpr int L4_MODEL = MAXB * L4_BW + MAXB * TW + 64;     // + 4 floats per collision point (sized at launch)
// per-lane LDS fields, one slot per step ([field][lane]); the fields from P_W on exist only in the all-LDS form of the kernel
constexpr int P_U = 0, P_DINV = 6 * L4_S, P_UU = 7 * L4_S, P_W = 8 * L4_S, P_N = 14 * L4_S, P_BH = 17 * L4_S, P_SIN = 18 * L4_S, P_COS = 19 * L4_S;
constexpr int l4_lds_floats(int num_points, bool regs) { return L4_STATE + L4_MODEL + num_points * 4 + (regs ? 8 : 20) * L4_S * 64; }

__device__ __forceinline__ float quad_from(float v, int src_lane_in_quad) { return __shfl(v, (threadIdx.x & ~3) | src_lane_in_quad, 64); }
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}
__device__ __forceinline__ void art_to_array(const ArtI& I, const Sp6& f, float* v) {
  const float t[27] = {I.A.xx, I.A.xy, I.A.xz, I.A.yy, I.A.yz, I.A.zz, I.B.m[0], I.B.m[1], I.B.m[2], I.B.m[3], I.B.m[4], I.B.m[5], I.B.m[6], I.B.m[7], I.B.m[8],
                       I.C.xx, I.C.xy, I.C.xz, I.C.yy, I.C.yz, I.C.zz, f.a.x, f.a.y, f.a.z, f.l.x, f.l.y, f.l.z};
#pragma unroll
  for (int i = 0; i < 27; ++i) v[i] = t[i];
}
__device__ __forceinline__ void art_add_array(ArtI& I, Sp6& f, const float* v) {
  I.A.xx += v[0]; I.A.xy += v[1]; I.A.xz += v[2]; I.A.yy += v[3]; I.A.yz += v[4]; I.A.zz += v[5];
#pragma unroll
  for (int i = 0; i < 9; ++i) I.B.m[i] += v[6 + i];
  I.C.xx += v[15]; I.C.xy += v[16]; I.C.xz += v[17]; I.C.yy += v[18]; I.C.yz += v[19]; I.C.zz += v[20];
  f.a.x += v[21]; f.a.y += v[22]; f.a.z += v[23]; f.l.x += v[24]; f.l.y += v[25]; f.l.z += v[26];
}

// rigid-body inertia and bias force of a body moving with (w, vl), gravity along -nz
__device__ __forceinline__ void body_inertia(const float* bc, V3 w, V3 vl, V3 nz, float gravity, ArtI& I, Sp6& p) {
  const float mass = bc[12];
  const V3 mc{bc[13], bc[14], bc[15]};
  I.A = {bc[16], bc[17], bc[18], bc[19], bc[20], bc[21]};
  I.B = {{0.f, -mc.z, mc.y, mc.z, 0.f, -mc.x, -mc.y, mc.x, 0.f}};
  I.C = {mass, 0.f, 0.f, mass, 0.f, mass};
  const V3 hn = mul(I.A, w) + cross(mc, vl);
  const V3 hl = mass * vl - cross(mc, w);
  const V3 gl = (-gravity) * nz;
  p.a = cross(w, hn) + cross(vl, hl) - cross(mc, gl);
  p.l = cross(w, hl) - mass * gl;
}

// ground contacts of one body's collision spheres, folded into its articulated inertia / bias force (linearly implicit)
__device__ __forceinline__ void body_contacts(const float* pts, int npt, int link, V3 w, V3 vl, V3 nz, float hk, float h, float kc, float cn, float mu,
                                              float veps, ArtI& I, Sp6& p, unsigned& touch) {
  for (int j = 0; j < npt; ++j) {
    const float* pt = pts + j * 4;
    const V3 r{pt[0], pt[1], pt[2]};
    const float d = pt[3] - (hk + dot(nz, r));
    if (d > 0.f) {
      touch |= 1u << link;
      const V3 vp = vl + cross(w, r);
      const float vn = dot(nz, vp);
      const V3 vt = vp - vn * nz;
      const float fn0 = fmaxf(kc * d - cn * vn, 0.f);
      const float ct = mu * fn0 / fmaxf(sqrtf(dot(vt, vt)), veps);
      const float fne = fmaxf(fn0 - h * kc * vn, 0.f);
      const V3 f = fne * nz - ct * vt;
      const float an = h * h * kc + h * cn - h * ct, at = h * ct;   // A = an nz nz^T + at 1
      const S3 Am{an * nz.x * nz.x + at, an * nz.x * nz.y, an * nz.x * nz.z, an * nz.y * nz.y + at, an * nz.y * nz.z, an * nz.z * nz.z + at};
      const M3 rA = skew_mul(r, full(Am));
      const M3 rArT = mul_skew(rA, V3{-r.x, -r.y, -r.z});
#pragma unroll
      for (int i = 0; i < 9; ++i) I.B.m[i] += rA.m[i];
      I.A.xx += rArT.m[0]; I.A.xy += rArT.m[1]; I.A.xz += rArT.m[2]; I.A.yy += rArT.m[4]; I.A.yz += rArT.m[5]; I.A.zz += rArT.m[8];
      I.C.xx += Am.xx; I.C.xy += Am.xy; I.C.xz += Am.xz; I.C.yy += Am.yy; I.C.yz += Am.yz; I.C.zz += Am.zz;
      p.a = p.a - cross(r, f);
      p.l = p.l - f;
    }
  }
}

template <bool REGS>
__global__ __launch_bounds__(64) void rigid_step4_kernel(addhip_rigid_model_t M, float* __restrict__ sim_pose, float* __restrict__ sim_vel,
                                                         const float* __restrict__ target, int tstride, int n, unsigned char* __restrict__ contact_flag,
                                                         unsigned* __restrict__ contact_bits) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x, e = lane >> 2, q = lane & 3;
  const int env0 = blockIdx.x * L4_ENVS;
  const int env = env0 + e;
  const int live = min(L4_ENVS, n - env0);
  float* st_pose = lds;
  float* st_vel = lds + L4_ENVS * 36;
  float* st_tgt = lds + L4_ENVS * 72;
  float* mbody = lds + L4_STATE;
  int* mtopo = reinterpret_cast<int*>(mbody + MAXB * L4_BW);
  int* mchain = mtopo + MAXB * TW;
  float* mpts = reinterpret_cast<float*>(mchain + 64);
  float* pl = mpts + M.num_points * 4;  // (the launch sizes the dynamic LDS for the model's own point count)
#define SP(c) st_pose[e * 36 + (c)]
#define SV(c) st_vel[e * 36 + (c)]
#define TG(c) st_tgt[e * 32 + (c)]
  // Per-body state handed from pass to pass, one slot per STEP of the pass: body velocity (w, vl), up-vector and height (pass 1 ->
  // passes 2 / 3), sin / cos of the joint angles (substep to substep), U, 1/D, u (pass 2 -> pass 3).  All of it can live in LDS
  // ([field][lane], 200 floats per lane: 69 KB per workgroup, two workgroups = two waves per CU).  REGS: the first 120 floats per lane are
  // VGPR arrays instead -- the step counter is wave-uniform, so they are registers addressed through M0 (s_set_gpr_idx), not memory --
  // which leaves 38 KB of LDS per workgroup: FOUR waves per CU, one per SIMD (65 536 envs: 1207 -> 648 us per control step).  The
  // indexed moves cost a lone wave 8 % (4096 envs: 150 -> 162 us), so launches that cannot fill two waves per CU keep the LDS form;
  // U, 1/D, u stay in LDS either way (with them the arrays pass the 256 directly addressable VGPRs and spill).
  float r_w0[L4_S], r_w1[L4_S], r_w2[L4_S], r_v0[L4_S], r_v1[L4_S], r_v2[L4_S], r_n0[L4_S], r_n1[L4_S], r_n2[L4_S], r_bh[L4_S], r_sin[L4_S], r_cos[L4_S];
#define PL(f, s_) pl[((f) + (s_)) * 64 + lane]
#define SLOT(reg, f, s_) (REGS ? reg[s_] : PL(f, s_))
  const int nb = M.num_bodies;
  for (int idx = lane; idx < live * 36; idx += 64) {
    st_pose[idx] = sim_pose[(size_t)env0 * 36 + idx];
    st_vel[idx] = sim_vel[(size_t)env0 * 36 + idx];
  }
  for (int idx = lane; idx < live * 32; idx += 64) {
    const int row = idx >> 5, col = idx & 31;
    st_tgt[idx] = col < 29 ? target[(size_t)(env0 + row) * tstride + col] : 0.f;
  }
  for (int idx = lane; idx < nb * BW; idx += 64) mbody[(idx / BW) * L4_BW + idx % BW] = M.body[idx];
  for (int idx = lane; idx < nb * TW; idx += 64) mtopo[idx] = M.topo[idx];
  mchain[lane] = M.chains[lane];
  for (int idx = lane; idx < M.num_points * 4; idx += 64) mpts[idx] = M.points[idx];
  __syncthreads();
  const bool on = e < live;  // quads past the last env run on a resting default state (kept in step, never written back)
  if (!on) {
    for (int c = q; c < 36; c += 4) { SP(c) = (c == 3) ? 1.f : (c == 2 ? 10.f : 0.f); SV(c) = 0.f; }
    for (int c = q; c < 32; c += 4) TG(c) = 0.f;
  }
  __syncthreads();

  const int* ch = mchain + q * 16;
  const int clen = ch[0], cstart = ch[1], cattach = ch[2];
  // steps at which a chain that hangs off another chain starts (pass 1 / 3 take the attach body's state there, pass 2 hands
  // the chain's inertia over one step earlier): wave-uniform bit masks
  unsigned hang_start = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (mchain[r * 16 + 2] >= 0 && mchain[r * 16] > 0) hang_start |= 1u << mchain[r * 16 + 1];

  const float h = M.dt / (float)M.substeps;
  const float kc = M.contact_stiffness, cn = M.contact_damping, veps = M.friction_vel_eps;
  const float gscale = (M.env_scale && on) ? M.env_scale[2 * env] : 1.f;
  const float mu = (M.env_scale && on) ? M.env_scale[2 * env + 1] : M.friction;
  unsigned touch = 0;

  V3 pos{SP(0), SP(1), SP(2)};
  float qw = SP(3), qx = SP(4), qy = SP(5), qz = SP(6);
  V3 vw{SV(0), SV(1), SV(2)}, ww{SV(3), SV(4), SV(5)};

  for (int sub = 0; sub < M.substeps; ++sub) {
    touch = 0;
    M3 R0 = {{1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
              2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
              2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)}};
    // ---------------- pass 1 (outward)
    const V3 rw = mulT(R0, ww), rv = mulT(R0, vw);
    const V3 rnz{R0.m[6], R0.m[7], R0.m[8]};
    const float rh = pos.z;
    V3 cw = rw, cv = rv, cnz = rnz;
    float chh = rh;
    for (int s = 0; s < L4_S; ++s) {
      if (hang_start & (1u << s)) {  // wave-uniform
        const int src = cattach >= 0 ? cattach : q;
        const float t[10] = {quad_from(cw.x, src), quad_from(cw.y, src), quad_from(cw.z, src), quad_from(cv.x, src), quad_from(cv.y, src),
                             quad_from(cv.z, src), quad_from(cnz.x, src), quad_from(cnz.y, src), quad_from(cnz.z, src), quad_from(chh, src)};
        if (cattach >= 0 && cstart == s) { cw = {t[0], t[1], t[2]}; cv = {t[3], t[4], t[5]}; cnz = {t[6], t[7], t[8]}; chh = t[9]; }
      }
      const int i = s - cstart;
      if (i >= 0 && i < clen) {
        const int k = ch[3 + i];
        const float* bc = mbody + k * L4_BW;
        const int* tp = mtopo + k * TW;
        const int ax = tp[1], dof = tp[2];
        const float qj = SP(7 + dof), qd = SV(6 + dof);
        float sn, cs;
        if (sub == 0) {
          sincosf(qj, &sn, &cs);
          SLOT(r_sin, P_SIN, s) = sn; SLOT(r_cos, P_COS, s) = cs;
        } else {
          sn = SLOT(r_sin, P_SIN, s); cs = SLOT(r_cos, P_COS, s);
        }
        const M3 R = joint_rot(bc + 3, ax, sn, cs);
        const V3 r{bc[0], bc[1], bc[2]};
        V3 w = mulT(R, cw);
        const V3 vl = mulT(R, cv - cross(r, cw));
        if (ax == 0) w.x += qd; else if (ax == 1) w.y += qd; else w.z += qd;
        const V3 nz = mulT(R, cnz);
        chh = chh + dot(cnz, r);
        SLOT(r_w0, P_W + 0 * L4_S, s) = w.x; SLOT(r_w1, P_W + 1 * L4_S, s) = w.y; SLOT(r_w2, P_W + 2 * L4_S, s) = w.z; SLOT(r_v0, P_W + 3 * L4_S, s) = vl.x; SLOT(r_v1, P_W + 4 * L4_S, s) = vl.y; SLOT(r_v2, P_W + 5 * L4_S, s) = vl.z;
        SLOT(r_n0, P_N + 0 * L4_S, s) = nz.x; SLOT(r_n1, P_N + 1 * L4_S, s) = nz.y; SLOT(r_n2, P_N + 2 * L4_S, s) = nz.z; SLOT(r_bh, P_BH, s) = chh;
        cw = w; cv = vl; cnz = nz;
      }
    }
    // ---------------- pass 2 (inward)
    float carry[27];  // this chain's articulated inertia and bias force in the frame of the body it was last handed to
#pragma unroll
    for (int i = 0; i < 27; ++i) carry[i] = 0.f;
    for (int s = L4_S - 1; s >= 0; --s) {
      const int i = s - cstart;
      const bool act = i >= 0 && i < clen;
      float hang[27];
#pragma unroll
      for (int x = 0; x < 27; ++x) hang[x] = 0.f;
      if (hang_start & (1u << (s + 1))) {  // a chain hanging off a body of this step has just finished: hand its inertia over
        for (int r = 0; r < 4; ++r) {
          const int ra = mchain[r * 16 + 2];
          if (ra >= 0 && mchain[r * 16 + 1] == s + 1 && mchain[r * 16] > 0) {  // wave-uniform
#pragma unroll
            for (int x = 0; x < 27; ++x) {
              const float t = quad_from(carry[x], r);
              if (q == ra) hang[x] += t;
            }
          }
        }
      }
      if (act) {
        const int k = ch[3 + i];
        const float* bc = mbody + k * L4_BW;
        const int* tp = mtopo + k * TW;
        const int ax = tp[1], dof = tp[2], pt0 = tp[5], npt = tp[6], link = tp[7];
        const V3 w{SLOT(r_w0, P_W + 0 * L4_S, s), SLOT(r_w1, P_W + 1 * L4_S, s), SLOT(r_w2, P_W + 2 * L4_S, s)}, vl{SLOT(r_v0, P_W + 3 * L4_S, s), SLOT(r_v1, P_W + 4 * L4_S, s), SLOT(r_v2, P_W + 5 * L4_S, s)};
        const V3 nz{SLOT(r_n0, P_N + 0 * L4_S, s), SLOT(r_n1, P_N + 1 * L4_S, s), SLOT(r_n2, P_N + 2 * L4_S, s)};
        const float hk = SLOT(r_bh, P_BH, s);
        ArtI I;
        Sp6 p;
        body_inertia(bc, w, vl, nz, M.gravity, I, p);
        if (i + 1 < clen) art_add_array(I, p, carry);  // the next body of this chain
        art_add_array(I, p, hang);                      // chains hanging off this body (zeros otherwise)
        if (hk < bc[29]) body_contacts(mpts + pt0 * 4, npt, link, w, vl, nz, hk, h, kc, cn, mu, veps, I, p, touch);
        // joint: PD torque with the implicit diagonal (stable PD unless the torque clamp is active), limit spring
        const float qj = SP(7 + dof), qd = SV(6 + dof);
        const float lo = bc[22], hi = bc[23], damp = bc[24], arm = bc[25], flim = fminf(bc[26], M.max_torque), kp = gscale * bc[27], kv = gscale * bc[28];
        const float tgt = fminf(fmaxf(TG(dof), lo + M.limit_margin), hi - M.limit_margin);
        const float tpd = kp * (tgt - qj) - kv * qd;
        float tau, dadd;
        if (fabsf(tpd) > flim) { tau = fminf(fmaxf(tpd, -flim), flim) - damp * qd; dadd = arm + h * damp; }
        else { tau = kp * (tgt - qj - h * qd) - (kv + damp) * qd; dadd = arm + h * (kv + damp) + h * h * kp; }
        if (qj < lo) { tau += M.limit_stiffness * (lo - qj - h * qd); dadd += h * h * M.limit_stiffness; }
        if (qj > hi) { tau += M.limit_stiffness * (hi - qj - h * qd); dadd += h * h * M.limit_stiffness; }
        V3 Ua, Ul;
        if (ax == 0) { Ua = {I.A.xx, I.A.xy, I.A.xz}; Ul = {I.B.m[0], I.B.m[1], I.B.m[2]}; }
        else if (ax == 1) { Ua = {I.A.xy, I.A.yy, I.A.yz}; Ul = {I.B.m[3], I.B.m[4], I.B.m[5]}; }
        else { Ua = {I.A.xz, I.A.yz, I.A.zz}; Ul = {I.B.m[6], I.B.m[7], I.B.m[8]}; }
        const float Dinv = 1.f / (comp(Ua, ax) + dadd);
        const float u = tau - comp(p.a, ax);
        PL(P_U, 6 * s) = Ua.x; PL(P_U, 6 * s + 1) = Ua.y; PL(P_U, 6 * s + 2) = Ua.z; PL(P_U, 6 * s + 3) = Ul.x; PL(P_U, 6 * s + 4) = Ul.y; PL(P_U, 6 * s + 5) = Ul.z;
        PL(P_DINV, s) = Dinv; PL(P_UU, s) = u;
        I.A.xx -= Dinv * Ua.x * Ua.x; I.A.xy -= Dinv * Ua.x * Ua.y; I.A.xz -= Dinv * Ua.x * Ua.z; I.A.yy -= Dinv * Ua.y * Ua.y; I.A.yz -= Dinv * Ua.y * Ua.z; I.A.zz -= Dinv * Ua.z * Ua.z;
        I.C.xx -= Dinv * Ul.x * Ul.x; I.C.xy -= Dinv * Ul.x * Ul.y; I.C.xz -= Dinv * Ul.x * Ul.z; I.C.yy -= Dinv * Ul.y * Ul.y; I.C.yz -= Dinv * Ul.y * Ul.z; I.C.zz -= Dinv * Ul.z * Ul.z;
        {
          const float ua[3] = {Ua.x, Ua.y, Ua.z}, ul[3] = {Ul.x, Ul.y, Ul.z};
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) I.B.m[3 * a + b] -= Dinv * ua[a] * ul[b];
        }
        const V3 ev = unit(ax);
        const V3 ca = qd * cross(w, ev), cl = qd * cross(vl, ev);
        const float ud = u * Dinv;
        p.a = p.a + mul(I.A, ca) + mul(I.B, cl) + ud * Ua;
        p.l = p.l + mulT(I.B, ca) + mul(I.C, cl) + ud * Ul;
        // to the parent's coordinates:  X^T Ia X,  X^T pa   (X = rot(R^T) xlt(r))
        const M3 R = joint_rot(bc + 3, ax, SLOT(r_sin, P_SIN, s), SLOT(r_cos, P_COS, s));
        const V3 r{bc[0], bc[1], bc[2]};
        ArtI P;
        P.A = rot_sym(R, I.A);
        P.C = rot_sym(R, I.C);
        const M3 Bp = matmul(matmul(R, I.B), transpose(R));
        const M3 rC = skew_mul(r, full(P.C));
        M3 Bn;
#pragma unroll
        for (int a = 0; a < 9; ++a) Bn.m[a] = Bp.m[a] + rC.m[a];
        const M3 t1 = skew_mul(r, transpose(Bn));
        const M3 t2 = mul_skew(Bp, r);
        P.A.xx += t1.m[0] - t2.m[0]; P.A.xy += t1.m[1] - t2.m[1]; P.A.xz += t1.m[2] - t2.m[2];
        P.A.yy += t1.m[4] - t2.m[4]; P.A.yz += t1.m[5] - t2.m[5]; P.A.zz += t1.m[8] - t2.m[8];
        P.B = Bn;
        Sp6 pp;
        pp.l = mul(R, p.l);
        pp.a = mul(R, p.a) + cross(r, pp.l);
        art_to_array(P, pp, carry);
      }
    }
    // ---------------- root: the chains hanging off it, its own inertia and contacts, a0 = -IA0^-1 pA0 (6x6 SPD, Cholesky)
    float a0[6];
    {
      ArtI I;
      Sp6 p;
      body_inertia(mbody, rw, rv, rnz, M.gravity, I, p);
      float tot[27];
#pragma unroll
      for (int x = 0; x < 27; ++x) tot[x] = quad_sum((cattach < 0 && clen > 0) ? carry[x] : 0.f);
      art_add_array(I, p, tot);
      if (rh < mbody[29]) body_contacts(mpts + mtopo[5] * 4, mtopo[6], mtopo[7], rw, rv, rnz, rh, h, kc, cn, mu, veps, I, p, touch);
      float A[6][6];
      A[0][0] = I.A.xx; A[0][1] = I.A.xy; A[0][2] = I.A.xz; A[1][1] = I.A.yy; A[1][2] = I.A.yz; A[2][2] = I.A.zz;
      A[3][3] = I.C.xx; A[3][4] = I.C.xy; A[3][5] = I.C.xz; A[4][4] = I.C.yy; A[4][5] = I.C.yz; A[5][5] = I.C.zz;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) A[a][3 + b] = I.B.m[3 * a + b];
      float b[6] = {-p.a.x, -p.a.y, -p.a.z, -p.l.x, -p.l.y, -p.l.z};
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int j = a; j < 6; ++j) {
          float sum = A[a][j];
#pragma unroll
          for (int t = 0; t < a; ++t) sum -= A[t][a] * A[t][j];
          A[a][j] = (j == a) ? sqrtf(fmaxf(sum, 1e-20f)) : sum / A[a][a];
        }
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        float sum = b[a];
#pragma unroll
        for (int t = 0; t < a; ++t) sum -= A[t][a] * b[t];
        b[a] = sum / A[a][a];
      }
#pragma unroll
      for (int a = 5; a >= 0; --a) {
        float sum = b[a];
#pragma unroll
        for (int t = a + 1; t < 6; ++t) sum -= A[a][t] * b[t];
        b[a] = sum / A[a][a];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) a0[a] = b[a];
    }
    // ---------------- pass 3 (outward): accelerations; joints are integrated as they are visited
    V3 caa{a0[0], a0[1], a0[2]}, cal{a0[3], a0[4], a0[5]};
    for (int s = 0; s < L4_S; ++s) {
      if (hang_start & (1u << s)) {
        const int src = cattach >= 0 ? cattach : q;
        const float t[6] = {quad_from(caa.x, src), quad_from(caa.y, src), quad_from(caa.z, src), quad_from(cal.x, src), quad_from(cal.y, src), quad_from(cal.z, src)};
        if (cattach >= 0 && cstart == s) { caa = {t[0], t[1], t[2]}; cal = {t[3], t[4], t[5]}; }
      }
      const int i = s - cstart;
      if (i >= 0 && i < clen) {
        const int k = ch[3 + i];
        const float* bc = mbody + k * L4_BW;
        const int* tp = mtopo + k * TW;
        const int ax = tp[1], dof = tp[2];
        const float s0 = SLOT(r_sin, P_SIN, s), c0 = SLOT(r_cos, P_COS, s);
        const M3 R = joint_rot(bc + 3, ax, s0, c0);
        const V3 r{bc[0], bc[1], bc[2]};
        const V3 w{SLOT(r_w0, P_W + 0 * L4_S, s), SLOT(r_w1, P_W + 1 * L4_S, s), SLOT(r_w2, P_W + 2 * L4_S, s)}, vl{SLOT(r_v0, P_W + 3 * L4_S, s), SLOT(r_v1, P_W + 4 * L4_S, s), SLOT(r_v2, P_W + 5 * L4_S, s)};
        const float qd = SV(6 + dof);
        const V3 ev = unit(ax);
        V3 aa = mulT(R, caa) + qd * cross(w, ev);
        const V3 al = mulT(R, cal - cross(r, caa)) + qd * cross(vl, ev);
        const V3 Ua{PL(P_U, 6 * s), PL(P_U, 6 * s + 1), PL(P_U, 6 * s + 2)}, Ul{PL(P_U, 6 * s + 3), PL(P_U, 6 * s + 4), PL(P_U, 6 * s + 5)};
        const float qdd = (PL(P_UU, s) - dot(Ua, aa) - dot(Ul, al)) * PL(P_DINV, s);
        if (ax == 0) aa.x += qdd; else if (ax == 1) aa.y += qdd; else aa.z += qdd;
        caa = aa; cal = al;
        const float qdn = qd + h * qdd;
        SV(6 + dof) = qdn;
        const float dq = h * qdn;
        SP(7 + dof) = SP(7 + dof) + dq;
        const float d2 = dq * dq;
        const float sd = dq * (1.f + d2 * (-1.f / 6.f + d2 * (1.f / 120.f - d2 * (1.f / 5040.f))));
        const float cd = 1.f + d2 * (-0.5f + d2 * (1.f / 24.f - d2 * (1.f / 720.f)));
        SLOT(r_sin, P_SIN, s) = s0 * cd + c0 * sd;
        SLOT(r_cos, P_COS, s) = c0 * cd - s0 * sd;
      }
    }
    // ---------------- root: semi-implicit Euler (all four lanes, identically)
    {
      const V3 aw = mul(R0, V3{a0[0], a0[1], a0[2]});
      const V3 al = mul(R0, V3{a0[3], a0[4], a0[5]}) + cross(ww, vw);
      ww = ww + h * aw;
      vw = vw + h * al;
      pos = pos + h * vw;
      const float wn = sqrtf(dot(ww, ww));
      const float half = 0.5f * h * wn;
      const float kq = wn > 1e-12f ? sinf(half) / wn : 0.5f * h;
      const float dw = cosf(half), dx = kq * ww.x, dy = kq * ww.y, dz = kq * ww.z;
      const float nw = dw * qw - dx * qx - dy * qy - dz * qz;
      const float nx = dw * qx + dx * qw + dy * qz - dz * qy;
      const float ny = dw * qy - dx * qz + dy * qw + dz * qx;
      const float nzq = dw * qz + dx * qy - dy * qx + dz * qw;
      const float inv = rsqrtf(nw * nw + nx * nx + ny * ny + nzq * nzq);
      qw = nw * inv; qx = nx * inv; qy = ny * inv; qz = nzq * inv;
    }
  }
  // ---- write back
  touch |= __shfl_xor(touch, 1, 64);
  touch |= __shfl_xor(touch, 2, 64);
  if (q == 0) {
    SP(0) = pos.x; SP(1) = pos.y; SP(2) = pos.z; SP(3) = qw; SP(4) = qx; SP(5) = qy; SP(6) = qz;
    SV(0) = vw.x; SV(1) = vw.y; SV(2) = vw.z; SV(3) = ww.x; SV(4) = ww.y; SV(5) = ww.z;
    SV(35) = 0.f;
    if (on) {
      if (contact_bits) contact_bits[env] = touch;
      if (contact_flag) contact_flag[env] = (touch & M.termination_mask) ? 1 : 0;
    }
  }
  __syncthreads();
  for (int idx = lane; idx < live * 36; idx += 64) {
    sim_pose[(size_t)env0 * 36 + idx] = st_pose[idx];
    sim_vel[(size_t)env0 * 36 + idx] = st_vel[idx];
  }
#undef SP
#undef SV
#undef TG
#undef PL
#undef SLOT
}


// ------------------------------------------------------------------ domain randomisation (build-defined extension; the reference has none)
// One launch per control step, BEFORE the physics step, driven by a device-resident control-step counter so that a captured
// hipGraph replays it unchanged: step index s = counter[0]; when due, env_scale[N,2] is redrawn (Philox stream (8<<40)+s) and the
// root's horizontal velocity is kicked (stream (9<<40)+s); one Philox call serves two envs.  The last workgroup to finish (ticket in
// counter[1]) leaves counter = {s+1, 0}: every workgroup has read s before it takes its ticket.
__global__ __launch_bounds__(256) void rigid_dr_kernel(addhip_rigid_dr_t d, float* env_scale, float* sim_vel, int n, unsigned long long* counter,
                                                       int advance) {
  const unsigned long long s = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool scale = advance ? (s > 0 && d.resample_interval > 0 && s % (unsigned)d.resample_interval == 0) : true;
  const bool push = advance && s > 0 && d.push_interval > 0 && s % (unsigned)d.push_interval == 0;
  if (scale || push) {
    const float gw = d.gain_hi - d.gain_lo, fw = d.friction_hi - d.friction_lo;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; 2 * q < n; q += (long long)gridDim.x * blockDim.x) {
      const long long e0 = 2 * q;
      const bool two = e0 + 1 < n;
      if (scale) {
        const U4 r = philox((uint64_t)q, (8ull << 40) + s, d.seed);
        env_scale[2 * e0 + 0] = d.gain_lo + u01(r.x) * gw;
        env_scale[2 * e0 + 1] = d.friction_lo + u01(r.y) * fw;
        if (two) {
          env_scale[2 * e0 + 2] = d.gain_lo + u01(r.z) * gw;
          env_scale[2 * e0 + 3] = d.friction_lo + u01(r.w) * fw;
        }
      }
      if (push) {
        const U4 r = philox((uint64_t)q, (9ull << 40) + s, d.seed);
        float* v0 = sim_vel + e0 * 36;
        v0[0] += (2.f * u01(r.x) - 1.f) * d.push_velocity;
        v0[1] += (2.f * u01(r.y) - 1.f) * d.push_velocity;
        if (two) {
          v0[36] += (2.f * u01(r.z) - 1.f) * d.push_velocity;
          v0[37] += (2.f * u01(r.w) - 1.f) * d.push_velocity;
        }
      }
    }
  }
  if (advance) {
    __syncthreads();  // every wave of this workgroup holds s
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned long long ticket = atomicAdd(counter + 1, 1ull);
      if (ticket == gridDim.x - 1) {
        __hip_atomic_store(counter, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(counter + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

}  // namespace

extern "C" int addhip_rigid_randomize(const addhip_rigid_dr_t* dr, float* env_scale, float* sim_vel, int32_t num_envs, uint64_t* step_counter,
                                      int32_t advance, void* stream) {
  ADDHIP_REQUIRE(dr && env_scale && sim_vel && step_counter && num_envs > 0, "rigid_randomize: bad arguments");
  ADDHIP_REQUIRE(dr->resample_interval >= 0 && dr->push_interval >= 0, "rigid_randomize: negative interval");
  ADDHIP_REQUIRE(dr->gain_lo > 0.f && dr->gain_hi >= dr->gain_lo && dr->friction_lo >= 0.f && dr->friction_hi >= dr->friction_lo, "rigid_randomize: bad ranges");
  ADDHIP_RECORDABLE(addhip_rigid_randomize, dr, env_scale, sim_vel, num_envs, step_counter, advance);
  long long g = ((num_envs + 1) / 2 + 255) / 256;
  if (g > 256) g = 256;
  hipLaunchKernelGGL(rigid_dr_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, *dr, env_scale, sim_vel, num_envs,
                     reinterpret_cast<unsigned long long*>(step_counter), (int)(advance != 0));
  return addhip::check_launch("rigid_dr_kernel");
}

// Per-device set-up done once (a control step is launch-latency scale at 4096 envs: no runtime queries on its path): the CU count that picks
// the kernel form, and the dynamic-LDS limits of the three kernels.  One slot per device ordinal, guarded by a mutex.
namespace {
struct DeviceSetup { bool done = false; int cus = 256; };
DeviceSetup* device_setup() {
  static std::mutex mu;
  static DeviceSetup slots[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  DeviceSetup& d = slots[dev];
  if (!d.done) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) d.cus = cus;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rigid_step4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * l4_lds_floats(MAXP, false))) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(rigid_step4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(float) * l4_lds_floats(MAXP, true))) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(rigid_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * LDS_FLOATS)) != hipSuccess)
      return nullptr;
    d.done = true;
  }
  return &d;
}
}  // namespace

extern "C" int addhip_rigid_step(const addhip_rigid_model_t* m, float* sim_pose, float* sim_vel, const float* target, int32_t target_stride,
                                 int32_t num_envs, uint8_t* contact_flag, uint32_t* contact_bits, void* stream) {
  ADDHIP_REQUIRE(m && sim_pose && sim_vel && target, "rigid_step: null argument");
  ADDHIP_REQUIRE(num_envs > 0 && target_stride >= ADDHIP_NUM_DOF, "rigid_step: bad sizes");
  ADDHIP_REQUIRE(m->num_bodies == ADDHIP_NUM_DOF + 1, "rigid_step: the packed state rows hold %d hinge dofs", ADDHIP_NUM_DOF);
  ADDHIP_REQUIRE(m->body && m->topo && (m->points || m->num_points == 0), "rigid_step: model tables missing");
  ADDHIP_REQUIRE(m->num_points >= 0 && m->num_points <= MAXP, "rigid_step: at most %d collision points", MAXP);
  ADDHIP_REQUIRE(m->substeps >= 1 && m->substeps <= 64 && m->dt > 0.f, "rigid_step: bad dt / substeps");
  ADDHIP_RECORDABLE(addhip_rigid_step, m, sim_pose, sim_vel, target, target_stride, num_envs, contact_flag, contact_bits);
  if (m->chains) {  // four lanes per environment
    // two waves per CU (the all-LDS form) serve up to 2 x 16 envs per CU; beyond that the register form runs four
    DeviceSetup* ds = device_setup();
    ADDHIP_REQUIRE(ds, "rigid_step: device set-up failed (hipFuncSetAttribute)");
    const bool regs = num_envs > 2 * L4_ENVS * ds->cus;
    const size_t shmem4 = sizeof(float) * l4_lds_floats(m->num_points, regs);  // G1 (301 points): 69 KB / 38.7 KB
    const dim3 grid4((num_envs + L4_ENVS - 1) / L4_ENVS);
    if (regs) hipLaunchKernelGGL(rigid_step4_kernel<true>, grid4, dim3(64), shmem4, (hipStream_t)stream, *m, sim_pose, sim_vel, target, target_stride, num_envs, contact_flag, contact_bits);
    else hipLaunchKernelGGL(rigid_step4_kernel<false>, grid4, dim3(64), shmem4, (hipStream_t)stream, *m, sim_pose, sim_vel, target, target_stride, num_envs, contact_flag, contact_bits);
    return addhip::check_launch("rigid_step4_kernel");
  }
  const size_t shmem = sizeof(float) * LDS_FLOATS;
  ADDHIP_REQUIRE(device_setup(), "rigid_step: device set-up failed (hipFuncSetAttribute)");
  hipLaunchKernelGGL(rigid_step_kernel, dim3((num_envs + WG - 1) / WG), dim3(WG), shmem, (hipStream_t)stream, *m, sim_pose, sim_vel, target,
                     target_stride, num_envs, contact_flag, contact_bits);
  return addhip::check_launch("rigid_step_kernel");
}
