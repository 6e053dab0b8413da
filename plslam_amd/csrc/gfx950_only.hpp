// gfx950_only.hpp -- force-included into every translation unit of libplslam_hip.so (plslam_amd/build.py: -include).
// The kernels are written for ONE target.  K1h / K1i issue their steady-state loads as LDS-DMA (global_load_lds_dword) through
// inline asm and wait for them with a hand-counted s_waitcnt vmcnt(2): correct under gfx9's in-order vmcnt and with at least two
// younger VMEM operations behind every request -- nothing the compiler can check; the matrix-core scans use the f8f6f4 MFMA,
// v_pk_minimum3_f16, v_permlane32_swap and v_bitop3_b32.  Another target must not compile them silently.
#pragma once
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libplslam_hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif
