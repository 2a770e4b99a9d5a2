// bf16 x 3 PLANE storage of fp32 values (ADDHIP_STORE_BF16X3; include/addhip.h, "plane storage").
//
// A fp32 value x is kept as three bf16 values hi + mid + lo == x EXACTLY (8 significant bits each, split by truncation: nothing is
// rounded; exact for every value whose lowest set bit is >= 2^-133, the resolution of a bf16 subnormal), so that a GEMM on stored operands forms the six products above 2^-24 |a||b| with the bf16 matrix cores at fp32-level error
// (gemm_x3.hip) WITHOUT re-splitting every operand tile in every workgroup that reads it (what gemm_split.hip does).  Layout of a row of
// `ld` values (ld % 8 == 0): groups of 8 consecutive values, each group = 48 bytes = [hi x 8][mid x 8][lo x 8]; element c of plane p is
// the u16 at index (c / 8) * 24 + p * 8 + (c % 8), row r starts at u16 index 3 * r * ld.  A 32-deep K stage of a row is then 192
// contiguous bytes whatever the plane, and one 16-byte chunk is 8 consecutive k of one plane = one lane's MFMA operand fragment.
#pragma once
#include <hip/hip_runtime.h>

namespace addhip_planes {

// x -> (hi, mid, lo) as fp32 bit patterns whose low 16 bits are zero (i.e. bf16 values in the high halves), exact
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid));  // <= 8 significant bits: already a bf16 value
}
// two bf16 (the high halves of a and b) -> one dword, a in the low half
__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// u16 index of element c of plane p inside its row
__device__ __forceinline__ size_t index(int c, int p) { return (size_t)(c >> 3) * 24 + (size_t)p * 8 + (size_t)(c & 7); }

// 8 consecutive values starting at column c (c % 8 == 0) of the row at `row` (u16*, 16-byte aligned): three 16-byte stores, 48 contiguous bytes
__device__ __forceinline__ void store8(unsigned short* row, int c, const float4& a, const float4& b) {
  unsigned h[8], m[8], l[8];
  split3(a.x, h[0], m[0], l[0]); split3(a.y, h[1], m[1], l[1]); split3(a.z, h[2], m[2], l[2]); split3(a.w, h[3], m[3], l[3]);
  split3(b.x, h[4], m[4], l[4]); split3(b.y, h[5], m[5], l[5]); split3(b.z, h[6], m[6], l[6]); split3(b.w, h[7], m[7], l[7]);
  uint4* dst = reinterpret_cast<uint4*>(row + (size_t)(c >> 3) * 24);
  dst[0] = make_uint4(pack2(h[0], h[1]), pack2(h[2], h[3]), pack2(h[4], h[5]), pack2(h[6], h[7]));
  dst[1] = make_uint4(pack2(m[0], m[1]), pack2(m[2], m[3]), pack2(m[4], m[5]), pack2(m[6], m[7]));
  dst[2] = make_uint4(pack2(l[0], l[1]), pack2(l[2], l[3]), pack2(l[4], l[5]), pack2(l[6], l[7]));
}
// 4 consecutive values starting at column c (c % 4 == 0): three 8-byte stores
__device__ __forceinline__ void store4(unsigned short* row, int c, const float4& v) {
  unsigned h[4], m[4], l[4];
  split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]); split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
  unsigned short* dst = row + (size_t)(c >> 3) * 24 + (c & 7);
  *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
  *reinterpret_cast<uint2*>(dst + 8) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
  *reinterpret_cast<uint2*>(dst + 16) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}
// the same on a FLAT buffer split in groups of 8 along its flat index i (i % 4 == 0): the parameter shadow
__device__ __forceinline__ void store4_flat(unsigned short* base, long long i, const float4& v) {
  unsigned h[4], m[4], l[4];
  split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]); split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
  unsigned short* dst = base + (i >> 3) * 24 + (i & 7);
  *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
  *reinterpret_cast<uint2*>(dst + 8) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
  *reinterpret_cast<uint2*>(dst + 16) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}
// one value
__device__ __forceinline__ void store1(unsigned short* row, int c, float v) {
  unsigned h, m, l;
  split3(v, h, m, l);
  unsigned short* dst = row + (size_t)(c >> 3) * 24 + (c & 7);
  dst[0] = (unsigned short)(h >> 16);
  dst[8] = (unsigned short)(m >> 16);
  dst[16] = (unsigned short)(l >> 16);
}

}  // namespace addhip_planes
