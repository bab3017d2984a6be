// tools/probe_fillers.hip -- what does one wave per SIMD pay for each kind of instruction, alone and in the shadow of an MFMA?
// One workgroup of 256 threads per CU (one wave per SIMD).  (a) N independent instructions of one kind back to back;
// (b) groups of [one v_mfma_f32_32x32x16_bf16 + n fillers of one kind], independent registers throughout.  Prints shader cycles
// per instruction / per group (s_memtime of wave 0).     hipcc -O3 --offload-arch=gfx950 tools/probe_fillers.hip -o probe_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND>
__global__ void __launch_bounds__(256) alone(unsigned long long* cyc, float* out, int iters) {
    float r = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) asm volatile(REP16("v_fma_f32 v40, v41, v42, v43\n v_fma_f32 v44, v45, v46, v47\n v_fma_f32 v48, v49, v50, v51\n v_fma_f32 v52, v53, v54, v55\n") ::: "v40", "v44", "v48", "v52");
        if constexpr (KIND == 1) asm volatile(REP16("v_exp_f32 v40, v41\n v_exp_f32 v44, v45\n v_exp_f32 v48, v49\n v_exp_f32 v52, v53\n") ::: "v40", "v44", "v48", "v52");
        if constexpr (KIND == 2) asm volatile(REP16("v_add_f32 v40, v41, v42\n v_add_f32 v44, v45, v46\n v_add_f32 v48, v49, v50\n v_add_f32 v52, v53, v54\n") ::: "v40", "v44", "v48", "v52");
        if constexpr (KIND == 3) asm volatile(REP16("v_cvt_pk_bf16_f32 v40, v41, v42\n v_cvt_pk_bf16_f32 v44, v45, v46\n v_cvt_pk_bf16_f32 v48, v49, v50\n v_cvt_pk_bf16_f32 v52, v53, v54\n") ::: "v40", "v44", "v48", "v52");
        if constexpr (KIND == 4) asm volatile(REP16("v_accvgpr_read_b32 v40, a1\n v_accvgpr_read_b32 v44, a2\n v_accvgpr_read_b32 v48, a3\n v_accvgpr_read_b32 v52, a4\n") ::: "v40", "v44", "v48", "v52");
        if constexpr (KIND == 5) asm volatile(REP16("v_add_f32 v40, v40, v42\n v_add_f32 v40, v40, v46\n v_add_f32 v40, v40, v50\n v_add_f32 v40, v40, v54\n") ::: "v40");   // dependent chain
        if constexpr (KIND == 6) asm volatile(REP16("v_exp_f32 v40, v41\n v_add_f32 v44, v45, v46\n v_add_f32 v48, v49, v50\n v_add_f32 v52, v53, v54\n") ::: "v40", "v44", "v48", "v52");   // 1 exp + 3 plain
        if constexpr (KIND == 7) asm volatile(REP16("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"));
        if constexpr (KIND == 8) asm volatile(REP16("v_mfma_f32_32x32x16_bf16 a[0:15], v[40:43], v[44:47], a[0:15]\n v_mfma_f32_32x32x16_bf16 a[16:31], v[40:43], v[44:47], a[16:31]\n v_mfma_f32_32x32x16_bf16 a[32:47], v[40:43], v[44:47], a[32:47]\n v_mfma_f32_32x32x16_bf16 a[48:63], v[40:43], v[44:47], a[48:63]\n") ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// group = 1 MFMA (rotating over 4 accumulators) + NF fillers of kind KIND
#define MF(a) "v_mfma_f32_32x32x16_bf16 a[" a "], v[40:43], v[44:47], a[" a "]\n"
template <int KIND, int NF>
__global__ void __launch_bounds__(256) shadow(unsigned long long* cyc, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 63) * 16;
    const unsigned ldsb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + 32768 + (threadIdx.x >> 6) * 1024));
    const unsigned vo = (threadIdx.x & 63) * 16;
    const __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(out, 0, 65536 * 4, 0x00020000);
    (void)la; (void)ldsb; (void)vo; (void)srd;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define FILL(reg)                                                                                             \
    if constexpr (KIND == 0) { if constexpr (NF > 0) asm volatile("v_fma_f32 v48, v49, v50, v51" ::: "v48"); if constexpr (NF > 1) asm volatile("v_fma_f32 v52, v53, v54, v55" ::: "v52"); if constexpr (NF > 2) asm volatile("v_fma_f32 v56, v57, v58, v59" ::: "v56"); if constexpr (NF > 3) asm volatile("v_fma_f32 v60, v61, v62, v63" ::: "v60"); if constexpr (NF > 4) asm volatile("v_fma_f32 v64, v65, v66, v67" ::: "v64"); if constexpr (NF > 5) asm volatile("v_fma_f32 v68, v69, v70, v71" ::: "v68"); if constexpr (NF > 6) asm volatile("v_fma_f32 v72, v73, v74, v75" ::: "v72"); if constexpr (NF > 7) asm volatile("v_fma_f32 v76, v77, v78, v79" ::: "v76"); } \
    if constexpr (KIND == 1) { if constexpr (NF > 0) asm volatile("v_exp_f32 v48, v49" ::: "v48"); if constexpr (NF > 1) asm volatile("v_exp_f32 v52, v53" ::: "v52"); if constexpr (NF > 2) asm volatile("v_exp_f32 v56, v57" ::: "v56"); if constexpr (NF > 3) asm volatile("v_exp_f32 v60, v61" ::: "v60"); if constexpr (NF > 4) asm volatile("v_exp_f32 v64, v65" ::: "v64"); if constexpr (NF > 5) asm volatile("v_exp_f32 v68, v69" ::: "v68"); if constexpr (NF > 6) asm volatile("v_exp_f32 v72, v73" ::: "v72"); if constexpr (NF > 7) asm volatile("v_exp_f32 v76, v77" ::: "v76"); } \
    if constexpr (KIND == 2) { if constexpr (NF > 0) asm volatile("v_add_f32 v48, v49, v50" ::: "v48"); if constexpr (NF > 1) asm volatile("v_add_f32 v52, v53, v54" ::: "v52"); if constexpr (NF > 2) asm volatile("v_add_f32 v56, v57, v58" ::: "v56"); if constexpr (NF > 3) asm volatile("v_add_f32 v60, v61, v62" ::: "v60"); if constexpr (NF > 4) asm volatile("v_add_f32 v64, v65, v66" ::: "v64"); if constexpr (NF > 5) asm volatile("v_add_f32 v68, v69, v70" ::: "v68"); if constexpr (NF > 6) asm volatile("v_add_f32 v72, v73, v74" ::: "v72"); if constexpr (NF > 7) asm volatile("v_add_f32 v76, v77, v78" ::: "v76"); } \
    if constexpr (KIND == 3) { if constexpr (NF > 0) asm volatile("v_cvt_pk_bf16_f32 v48, v49, v50" ::: "v48"); if constexpr (NF > 1) asm volatile("v_cvt_pk_bf16_f32 v52, v53, v54" ::: "v52"); if constexpr (NF > 2) asm volatile("v_cvt_pk_bf16_f32 v56, v57, v58" ::: "v56"); if constexpr (NF > 3) asm volatile("v_cvt_pk_bf16_f32 v60, v61, v62" ::: "v60"); if constexpr (NF > 4) asm volatile("v_cvt_pk_bf16_f32 v64, v65, v66" ::: "v64"); if constexpr (NF > 5) asm volatile("v_cvt_pk_bf16_f32 v68, v69, v70" ::: "v68"); if constexpr (NF > 6) asm volatile("v_cvt_pk_bf16_f32 v72, v73, v74" ::: "v72"); if constexpr (NF > 7) asm volatile("v_cvt_pk_bf16_f32 v76, v77, v78" ::: "v76"); } \
    if constexpr (KIND == 4) { if constexpr (NF > 0) asm volatile("v_fma_f32 v48, v49, s4, v51" ::: "v48"); if constexpr (NF > 1) asm volatile("v_exp_f32 v52, v53" ::: "v52"); if constexpr (NF > 2) asm volatile("v_add_f32 v56, v57, v58" ::: "v56"); if constexpr (NF > 3) asm volatile("v_fma_f32 v60, v61, s4, v63" ::: "v60"); if constexpr (NF > 4) asm volatile("v_exp_f32 v64, v65" ::: "v64"); if constexpr (NF > 5) asm volatile("v_add_f32 v68, v69, v70" ::: "v68"); if constexpr (NF > 6) asm volatile("v_cvt_pk_bf16_f32 v72, v73, v74" ::: "v72"); if constexpr (NF > 7) asm volatile("v_fma_f32 v76, v77, s4, v79" ::: "v76"); }
#define LDS1(i) if constexpr (NF > i) asm volatile("ds_read_b64_tr_b16 v[%c1:%c2], %0 offset:%c3" :: "v"(la), "n"(48 + 2 * i), "n"(49 + 2 * i), "n"(256 * i) : "memory");
#define LDS2(i) if constexpr (NF > i) asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" :: "v"(la), "n"(128 + 4 * i), "n"(131 + 4 * i), "n"(512 * i) : "memory");
#define LDS3(i) if constexpr (NF > i) asm volatile("ds_read_b128 v[%c1:%c2], %0 offset:%c3" :: "v"(la), "n"(128 + 4 * i), "n"(131 + 4 * i), "n"(512 * i) : "memory");
#define LDS4(i) if constexpr (NF > i) asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%c3" :: "v"(la), "n"(128 + 2 * i), "n"(129 + 2 * i), "n"(256 * i) : "memory");
#define GROUP(a) asm volatile(MF(a) ::: "memory"); FILL(0) \
    if constexpr (KIND == 11) { if constexpr (NF > 0) asm volatile("v_pk_fma_f32 v[48:49], v[50:51], s[4:5], v[52:53] op_sel_hi:[1,1,0]" ::: "v48", "v49"); if constexpr (NF > 1) asm volatile("v_pk_fma_f32 v[56:57], v[58:59], s[4:5], v[60:61] op_sel_hi:[1,1,0]" ::: "v56", "v57"); if constexpr (NF > 2) asm volatile("v_pk_fma_f32 v[64:65], v[66:67], s[4:5], v[68:69] op_sel_hi:[1,1,0]" ::: "v64", "v65"); if constexpr (NF > 3) asm volatile("v_pk_fma_f32 v[72:73], v[74:75], s[4:5], v[76:77] op_sel_hi:[1,1,0]" ::: "v72", "v73"); if constexpr (NF > 4) asm volatile("v_pk_fma_f32 v[80:81], v[82:83], s[4:5], v[84:85] op_sel_hi:[1,1,0]" ::: "v80", "v81"); if constexpr (NF > 5) asm volatile("v_pk_fma_f32 v[88:89], v[90:91], s[4:5], v[92:93] op_sel_hi:[1,1,0]" ::: "v88", "v89"); if constexpr (NF > 6) asm volatile("v_pk_fma_f32 v[96:97], v[98:99], s[4:5], v[100:101] op_sel_hi:[1,1,0]" ::: "v96", "v97"); if constexpr (NF > 7) asm volatile("v_pk_fma_f32 v[104:105], v[106:107], s[4:5], v[108:109] op_sel_hi:[1,1,0]" ::: "v104", "v105"); } \
    if constexpr (KIND == 12) { if constexpr (NF > 0) asm volatile("v_pk_add_f32 v[48:49], v[50:51], v[52:53]" ::: "v48", "v49"); if constexpr (NF > 1) asm volatile("v_pk_add_f32 v[56:57], v[58:59], v[60:61]" ::: "v56", "v57"); if constexpr (NF > 2) asm volatile("v_pk_add_f32 v[64:65], v[66:67], v[68:69]" ::: "v64", "v65"); if constexpr (NF > 3) asm volatile("v_pk_add_f32 v[72:73], v[74:75], v[76:77]" ::: "v72", "v73"); if constexpr (NF > 4) asm volatile("v_pk_add_f32 v[80:81], v[82:83], v[84:85]" ::: "v80", "v81"); if constexpr (NF > 5) asm volatile("v_pk_add_f32 v[88:89], v[90:91], v[92:93]" ::: "v88", "v89"); if constexpr (NF > 6) asm volatile("v_pk_add_f32 v[96:97], v[98:99], v[100:101]" ::: "v96", "v97"); if constexpr (NF > 7) asm volatile("v_pk_add_f32 v[104:105], v[106:107], v[108:109]" ::: "v104", "v105"); } \
    if constexpr (KIND == 13) { if constexpr (NF > 0) asm volatile("v_pk_mul_f32 v[48:49], v[50:51], v[52:53]" ::: "v48", "v49"); if constexpr (NF > 1) asm volatile("v_pk_mul_f32 v[56:57], v[58:59], v[60:61]" ::: "v56", "v57"); if constexpr (NF > 2) asm volatile("v_pk_mul_f32 v[64:65], v[66:67], v[68:69]" ::: "v64", "v65"); if constexpr (NF > 3) asm volatile("v_pk_mul_f32 v[72:73], v[74:75], v[76:77]" ::: "v72", "v73"); if constexpr (NF > 4) asm volatile("v_pk_mul_f32 v[80:81], v[82:83], v[84:85]" ::: "v80", "v81"); if constexpr (NF > 5) asm volatile("v_pk_mul_f32 v[88:89], v[90:91], v[92:93]" ::: "v88", "v89"); if constexpr (NF > 6) asm volatile("v_pk_mul_f32 v[96:97], v[98:99], v[100:101]" ::: "v96", "v97"); if constexpr (NF > 7) asm volatile("v_pk_mul_f32 v[104:105], v[106:107], v[108:109]" ::: "v104", "v105"); } \
    if constexpr (KIND == 14) { if constexpr (NF > 0) asm volatile("v_pk_fma_f32 v[48:49], v[50:51], s[4:5], v[52:53] op_sel_hi:[1,1,0]" ::: "v48", "v49"); if constexpr (NF > 1) asm volatile("v_exp_f32 v56, v57" ::: "v56"); if constexpr (NF > 2) asm volatile("v_exp_f32 v58, v59" ::: "v58"); if constexpr (NF > 3) asm volatile("v_pk_add_f32 v[60:61], v[60:61], v[62:63]" ::: "v60", "v61"); if constexpr (NF > 4) asm volatile("v_cvt_pk_bf16_f32 v64, v65, v66" ::: "v64"); if constexpr (NF > 5) asm volatile("v_pk_fma_f32 v[68:69], v[70:71], s[4:5], v[72:73] op_sel_hi:[1,1,0]" ::: "v68", "v69"); if constexpr (NF > 6) asm volatile("v_exp_f32 v74, v75" ::: "v74"); if constexpr (NF > 7) asm volatile("v_exp_f32 v76, v77" ::: "v76"); } \
    if constexpr (KIND == 15) { if constexpr (NF > 0) asm volatile("v_dot2c_f32_f16 v48, 0x3c003c00, v49" ::: "v48"); if constexpr (NF > 1) asm volatile("v_dot2c_f32_f16 v56, 0x3c003c00, v57" ::: "v56"); if constexpr (NF > 2) asm volatile("v_dot2c_f32_f16 v64, 0x3c003c00, v65" ::: "v64"); if constexpr (NF > 3) asm volatile("v_dot2c_f32_f16 v72, 0x3c003c00, v73" ::: "v72"); if constexpr (NF > 4) asm volatile("v_dot2c_f32_f16 v80, 0x3c003c00, v81" ::: "v80"); if constexpr (NF > 5) asm volatile("v_dot2c_f32_f16 v88, 0x3c003c00, v89" ::: "v88"); if constexpr (NF > 6) asm volatile("v_dot2c_f32_f16 v96, 0x3c003c00, v97" ::: "v96"); if constexpr (NF > 7) asm volatile("v_dot2c_f32_f16 v104, 0x3c003c00, v105" ::: "v104"); } \
    if constexpr (KIND == 16) { if constexpr (NF > 0) asm volatile("v_dot2c_f32_f16 v48, v50, v49" ::: "v48"); if constexpr (NF > 1) asm volatile("v_dot2c_f32_f16 v56, v58, v57" ::: "v56"); if constexpr (NF > 2) asm volatile("v_dot2c_f32_f16 v64, v66, v65" ::: "v64"); if constexpr (NF > 3) asm volatile("v_dot2c_f32_f16 v72, v74, v73" ::: "v72"); if constexpr (NF > 4) asm volatile("v_dot2c_f32_f16 v80, v82, v81" ::: "v80"); if constexpr (NF > 5) asm volatile("v_dot2c_f32_f16 v88, v90, v89" ::: "v88"); if constexpr (NF > 6) asm volatile("v_dot2c_f32_f16 v96, v98, v97" ::: "v96"); if constexpr (NF > 7) asm volatile("v_dot2c_f32_f16 v104, v106, v105" ::: "v104"); } \
    if constexpr (KIND == 17) { if constexpr (NF > 0) asm volatile("v_dot2_f32_f16 v48, v49, v50, v48" ::: "v48"); if constexpr (NF > 1) asm volatile("v_dot2_f32_f16 v56, v57, v58, v56" ::: "v56"); if constexpr (NF > 2) asm volatile("v_dot2_f32_f16 v64, v65, v66, v64" ::: "v64"); if constexpr (NF > 3) asm volatile("v_dot2_f32_f16 v72, v73, v74, v72" ::: "v72"); if constexpr (NF > 4) asm volatile("v_dot2_f32_f16 v80, v81, v82, v80" ::: "v80"); if constexpr (NF > 5) asm volatile("v_dot2_f32_f16 v88, v89, v90, v88" ::: "v88"); if constexpr (NF > 6) asm volatile("v_dot2_f32_f16 v96, v97, v98, v96" ::: "v96"); if constexpr (NF > 7) asm volatile("v_dot2_f32_f16 v104, v105, v106, v104" ::: "v104"); } \
    if constexpr (KIND == 18) { if constexpr (NF > 0) asm volatile("v_dot2c_f32_bf16 v48, v50, v49" ::: "v48"); if constexpr (NF > 1) asm volatile("v_dot2c_f32_bf16 v56, v58, v57" ::: "v56"); if constexpr (NF > 2) asm volatile("v_dot2c_f32_bf16 v64, v66, v65" ::: "v64"); if constexpr (NF > 3) asm volatile("v_dot2c_f32_bf16 v72, v74, v73" ::: "v72"); if constexpr (NF > 4) asm volatile("v_dot2c_f32_bf16 v80, v82, v81" ::: "v80"); if constexpr (NF > 5) asm volatile("v_dot2c_f32_bf16 v88, v90, v89" ::: "v88"); if constexpr (NF > 6) asm volatile("v_dot2c_f32_bf16 v96, v98, v97" ::: "v96"); if constexpr (NF > 7) asm volatile("v_dot2c_f32_bf16 v104, v106, v105" ::: "v104"); } \
    if constexpr (KIND == 5) { LDS1(0) LDS1(1) LDS1(2) LDS1(3) LDS1(4) LDS1(5) LDS1(6) LDS1(7) } \
    if constexpr (KIND == 6) { LDS2(0) LDS2(1) LDS2(2) LDS2(3) LDS2(4) LDS2(5) LDS2(6) LDS2(7) } \
    if constexpr (KIND == 9) { LDS3(0) LDS3(1) LDS3(2) LDS3(3) LDS3(4) LDS3(5) LDS3(6) LDS3(7) } \
    if constexpr (KIND == 10) { LDS4(0) LDS4(1) LDS4(2) LDS4(3) LDS4(4) LDS4(5) LDS4(6) LDS4(7) } \
    if constexpr (KIND == 7) { asm volatile("v_exp_f32 v52, v53" ::: "v52"); asm volatile("v_fma_f32 v48, v49, s4, v51" ::: "v48"); asm volatile("v_add_f32 v56, v57, v58" ::: "v56"); if constexpr (NF > 0) asm volatile("v_cvt_pk_bf16_f32 v72, v73, v74" ::: "v72"); LDS1(1) LDS1(2) if constexpr (NF > 3) asm volatile("v_add_f32 v60, v61, v62" ::: "v60"); if constexpr (NF > 4) asm volatile("v_add_f32 v64, v61, v62" ::: "v64");} \
    if constexpr (KIND == 8) { if constexpr (NF > 0) asm volatile("s_add_u32 m0, %0, 0\n v_add_f32 v60, v61, v62\n buffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(ldsb), "v"(vo), "s"(srd) : "memory", "m0", "scc", "v60"); if constexpr (NF > 1) asm volatile("v_exp_f32 v52, v53" ::: "v52"); if constexpr (NF > 2) asm volatile("v_fma_f32 v48, v49, s4, v51" ::: "v48"); }
        REP4(GROUP("64:79") GROUP("80:95") GROUP("96:111") GROUP("112:127"))
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static unsigned long long* cyc; static float* out;
template <class K> static double run(K k, int iters) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, cyc, out, iters); hipDeviceSynchronize(); }
    unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); return (double)c;
}
template <int KIND> static void sweep(const char* name) {
    const int it = 500;
    printf("%-14s per [MFMA + n]: n=0 %5.1f  1 %5.1f  2 %5.1f  3 %5.1f  4 %5.1f  5 %5.1f  6 %5.1f  7 %5.1f  8 %5.1f\n", name,
           run(shadow<KIND, 0>, it) / (16.0 * it), run(shadow<KIND, 1>, it) / (16.0 * it), run(shadow<KIND, 2>, it) / (16.0 * it),
           run(shadow<KIND, 3>, it) / (16.0 * it), run(shadow<KIND, 4>, it) / (16.0 * it), run(shadow<KIND, 5>, it) / (16.0 * it),
           run(shadow<KIND, 6>, it) / (16.0 * it), run(shadow<KIND, 7>, it) / (16.0 * it), run(shadow<KIND, 8>, it) / (16.0 * it));
}
int main() {
    hipMalloc(&cyc, 64); hipMalloc(&out, 256 * 256 * 4);
    const int it = 500;
    const char* names[] = {"v_fma_f32", "v_exp_f32", "v_add_f32", "v_cvt_pk_bf16", "v_accvgpr_read", "v_add dependent", "1 exp + 3 add", "s_nop 0", "mfma 32x32x16"};
    double r[9] = {run(alone<0>, it), run(alone<1>, it), run(alone<2>, it), run(alone<3>, it), run(alone<4>, it), run(alone<5>, it), run(alone<6>, it), run(alone<7>, it), run(alone<8>, it)};
    for (int i = 0; i < 9; ++i) printf("alone  %-16s %6.2f cycles per instruction\n", names[i], r[i] / (64.0 * it));
    sweep<0>("v_fma_f32"); sweep<1>("v_exp_f32"); sweep<2>("v_add_f32"); sweep<3>("v_cvt_pk_bf16"); sweep<4>("softmax mix");
    sweep<5>("ds_read_tr_b64"); sweep<6>("ds_read_b128>a"); sweep<7>("exp+fma+add +n"); sweep<8>("dma,exp,fma"); sweep<9>("ds_read_b128>v"); sweep<10>("ds_read_tr>a");
    sweep<11>("v_pk_fma_f32"); sweep<12>("v_pk_add_f32"); sweep<13>("v_pk_mul_f32"); sweep<14>("pk softmax mix");
    sweep<15>("dot2c_f16 lit"); sweep<16>("dot2c_f16 vgpr"); sweep<17>("dot2_f16 vop3p"); sweep<18>("dot2c_bf16");
    return 0;
}
