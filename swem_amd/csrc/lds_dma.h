// LDS-DMA helpers shared by the convolution kernels (forward / data gradient: conv.hip; weight gradient: train_conv.hip).
#pragma once
#include <hip/hip_runtime.h>

// 16 bytes per lane, global/L2 -> LDS without a register stop (buffer_load_dwordx4 ... lds): the 64 lanes of a wave land
// in 64 consecutive 16-byte slots starting at the wave-uniform LDS address in M0; out-of-range offsets land as zeros.
// Inline asm on purpose: with the builtin, hipcc counts the transfer as an LDS write of unknown address and puts
// s_waitcnt vmcnt(0) in front of the next ds_read, which drains the ring every k-block; here the only waits are the
// counted ones in the kernel.  M0 is saved and restored (compiler-reserved); s_nop 4 covers a descriptor or offset SGPR
// written just before.
typedef int i32x4 __attribute__((ext_vector_type(4)));
// raw buffer descriptor: base, stride 0, byte range, raw 32-bit data format
__device__ __forceinline__ i32x4 raw_rsrc(const void *base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)(unsigned)((a >> 32) & 0xffffu);
  r[2] = (int)bytes;
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ void dma16(i32x4 r4, unsigned lds_addr, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the kernel's launch stub
  // M0 is written in the statement that consumes it and is not kept live by hipcc anywhere in these kernels (no other
  // M0 user: checked in the ISA), so it is not saved; the one wait state between the M0 write and the transfer is the
  // s_nop.  Descriptor and offset SGPRs are written by scalar instructions only (no VALU->SGPR hazard).
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, %3 offen lds"
               :
               : "v"(voff), "s"(lds_addr), "s"(r4), "s"(soff)
               : "memory");
#endif
}

