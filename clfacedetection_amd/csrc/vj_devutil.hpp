// Small device-side helpers shared by the gfx950 kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vj {

// Read-only, wave-uniform data goes through address space 4 so that the compiler may
// use s_load (scalar cache) even though the kernel also stores to global memory.
template <typename T>
using kptr = const T __attribute__((address_space(4)))*;
template <typename T>
__device__ __forceinline__ kptr<T> as_k(const T* p) {
    return (kptr<T>)(uintptr_t)p;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Number of set bits of `mask` below this lane.
__device__ __forceinline__ uint32_t mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Image gathers go through buffer loads: the 128-bit resource descriptor and the
// scalar offset (a rectangle corner, uniform across the wave) live in SGPRs and the
// lane contributes only its 32-bit window offset, so a gather costs no VALU address
// arithmetic at all:  buffer_load_dword v, v_off, s[rsrc], s_corner offen.
// Out-of-range offsets return 0 instead of faulting.
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t ld_u32(rsrc_t r, uint32_t lane_off, uint32_t uniform_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, uniform_off, 0);
}
__device__ __forceinline__ uint64_t ld_u64(rsrc_t r, uint32_t lane_off, uint32_t uniform_off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, uniform_off, 0);
    return (uint64_t)v[0] | ((uint64_t)v[1] << 32);
}

// 64-byte node record as the scalar unit loads it (one s_load_dwordx16).
typedef uint32_t NodeRecDev __attribute__((ext_vector_type(16)));

}  // namespace vj
