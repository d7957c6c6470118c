#!/usr/bin/env python3
"""Issue rates of gfx950 vector instructions, measured.

Generates a HIP program in which every wavefront runs a loop of 16 copies of a
short instruction pattern (inline assembly, independent registers unless the
pattern says otherwise), launches it with 1, 2, 3, 4, 5 and 8 wavefronts per
SIMD and prints the cycles one SIMD spends per wave-instruction.  What the
path kernels are tuned by came out of this table:

  * v_fma / v_add / v_mul / v_sub _f32, v_mov, v_and / v_xor / shifts / v_add_u32
    issue at about 3.0 cycles per wavefront once several wavefronts share the SIMD;
  * v_min / v_max / v_max3 / v_med3 _f32, every v_cmp, v_cndmask, every v_pk_*_f32,
    conversions, integer multiplies, v_lshl_add_u32, all f64 arithmetic and the
    v_div_scale / v_div_fmas / v_div_fixup of an IEEE division at about 4.5-5.2,
    and a mix of the two classes runs faster than the sum of its parts (an
    instruction of the first class issues beside one of the second);
  * v_sqrt / v_rcp / v_log / v_sin _f32 at about 9.2;
  * a v_cndmask_b32 in its short encoding (condition in vcc) that directly follows
    another one takes about 17 cycles; the long encoding does not.

    python tools/valu_rate.py            # writes and builds tools/valu_rate_bin (no GPU needed)
    ./tools/valu_rate_bin                # on the GPU

Cycles are counted at the clock the device properties report (2.4 GHz); the
sustained clock is lower, so the numbers are for comparing instructions with
each other (results of one run: profiles/r02_valu_issue_rates.txt).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))

# (label, instructions, counted instructions per copy, clobbers)
# operands: %0 = v[i] (read-write), %1 = a, %2 = b, %3 = v[i+5], %4 = v[i+9], %5 = packed pair p[i], %6 = double d[i]
KINDS = [
    ("v_fma_f32", "v_fma_f32 %0, %0, %1, %2", 1, ""),
    ("v_fma_f32, dependent chain", "v_fma_f32 %7, %7, %1, %2", 1, ""),
    ("v_fmac_f32", "v_fmac_f32 %0, %1, %2", 1, ""),
    ("v_add_f32", "v_add_f32 %0, %0, %1", 1, ""),
    ("v_sub_f32", "v_sub_f32 %0, %0, %1", 1, ""),
    ("v_mul_f32", "v_mul_f32 %0, %0, %1", 1, ""),
    ("v_min_f32", "v_min_f32 %0, %0, %1", 1, ""),
    ("v_max_f32", "v_max_f32 %0, %0, %1", 1, ""),
    ("v_max3_f32", "v_max3_f32 %0, %0, %1, %2", 1, ""),
    ("v_med3_f32", "v_med3_f32 %0, %0, %1, %2", 1, ""),
    ("v_pk_fma_f32", "v_pk_fma_f32 %5, %5, %5, %5", 1, ""),
    ("v_pk_mul_f32", "v_pk_mul_f32 %5, %5, %5", 1, ""),
    ("v_pk_add_f32", "v_pk_add_f32 %5, %5, %5", 1, ""),
    ("v_mov_b32", "v_mov_b32 %0, %3", 1, ""),
    ("v_and_b32", "v_and_b32 %0, %0, %1", 1, ""),
    ("v_xor_b32", "v_xor_b32 %0, %0, %1", 1, ""),
    ("v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %0", 1, ""),
    ("v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %0", 1, ""),
    ("v_add_u32", "v_add_u32 %0, %0, %1", 1, ""),
    ("v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 3, %1", 1, ""),
    ("v_xad_u32", "v_xad_u32 %0, %0, %1, %2", 1, ""),
    ("v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2", 1, ""),
    ("v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", 1, ""),
    ("v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1", 1, ""),
    ("v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1", 1, ""),
    ("v_bfe_u32", "v_bfe_u32 %0, %0, 3, 5", 1, ""),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, -1, %0", 1, ""),
    ("v_readlane_b32", "v_readlane_b32 s22, %0, 5", 1, "s22"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0", 1, ""),
    ("v_sqrt_f32", "v_sqrt_f32 %0, %0", 1, ""),
    ("v_rcp_f32", "v_rcp_f32 %0, %0", 1, ""),
    ("v_log_f32", "v_log_f32 %0, %0", 1, ""),
    ("v_sin_f32", "v_sin_f32 %0, %0", 1, ""),
    ("v_fma_f64", "v_fma_f64 %6, %6, %6, %6", 1, ""),
    ("v_mul_f64", "v_mul_f64 %6, %6, %6", 1, ""),
    ("v_add_f64", "v_add_f64 %6, %6, %6", 1, ""),
    ("v_cvt_f64_f32", "v_cvt_f64_f32 %6, %0", 1, ""),
    ("v_cvt_f32_f64", "v_cvt_f32_f64 %0, %6", 1, ""),
    ("v_div_scale_f32", "v_div_scale_f32 %0, vcc, %0, %1, %0", 1, "vcc"),
    ("v_div_fmas_f32", "v_div_fmas_f32 %0, %0, %1, %2", 1, ""),
    ("v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %1, %2", 1, ""),
    ("v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 vcc, %0, %1", 1, "vcc"),
    ("v_cmp_lt_f32 -> s[20:21]", "v_cmp_lt_f32 s[20:21], %0, %1", 1, 's20", "s21'),
    ("v_cndmask_b32_e64, s[20:21]", "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", 1, ""),
    ("v_cndmask_b32_e64, vcc", "v_cndmask_b32_e64 %0, %0, %1, vcc", 1, ""),
    ("v_cndmask_b32_e32 (vcc), back to back", "v_cndmask_b32_e32 %0, %0, %1, vcc", 1, ""),
    ("mix: v_fma_f32 + v_max3_f32", "v_fma_f32 %0, %0, %1, %2\\n v_max3_f32 %3, %3, %1, %2", 2, ""),
    ("mix: v_max3_f32 + 3 v_fma_f32", "v_max3_f32 %0, %0, %1, %2\\n v_fma_f32 %3, %3, %1, %2\\n v_fma_f32 %4, %4, %1, %2\\n v_fma_f32 %3, %3, %1, %2", 4, ""),
    ("mix: v_pk_fma_f32 + v_fma_f32", "v_pk_fma_f32 %5, %5, %5, %5\\n v_fma_f32 %3, %3, %1, %2", 2, ""),
    ("v_cmp vcc; s_nop 1; v_cndmask_e32", "v_cmp_lt_f32 vcc, %0, %1\\n s_nop 1\\n v_cndmask_b32_e32 %0, %0, %2, vcc", 2, "vcc"),
    ("v_cmp sgpr; s_nop 1; v_cndmask_e64", "v_cmp_lt_f32 s[20:21], %0, %1\\n s_nop 1\\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]", 2, 's20", "s21'),
    ("v_cmp vcc; s_nop 1; 3 v_cndmask_e32", "v_cmp_lt_f32 vcc, %0, %1\\n s_nop 1\\n v_cndmask_b32_e32 %0, %0, %2, vcc\\n v_cndmask_b32_e32 %3, %3, %2, vcc\\n v_cndmask_b32_e32 %4, %4, %2, vcc", 4, "vcc"),
    ("v_cmp sgpr; s_nop 1; 3 v_cndmask_e64", "v_cmp_lt_f32 s[20:21], %0, %1\\n s_nop 1\\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\\n v_cndmask_b32_e64 %3, %3, %2, s[20:21]\\n v_cndmask_b32_e64 %4, %4, %2, s[20:21]", 4, 's20", "s21'),
    ("v_cmp vcc; s_nop 1; v_cndmask_e32; v_fma; v_cndmask_e32", "v_cmp_lt_f32 vcc, %0, %1\\n s_nop 1\\n v_cndmask_b32_e32 %0, %0, %2, vcc\\n v_fma_f32 %4, %4, %2, %1\\n v_cndmask_b32_e32 %3, %3, %2, vcc", 4, "vcc"),
]

HEAD = r'''// generated by tools/valu_rate.py
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kIters = 4096;
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, float seed) {
    float v[16];
    double d[16];
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { v[i] = seed + i + threadIdx.x; d[i] = v[i]; p[i] = (f2)(v[i], v[i] + 1.f); }
    float a = seed * 1.0001f, b = seed * 0.5f;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_mov_b64 s[20:21], vcc" : : "v"(a), "v"(v[3]) : "vcc", "s20", "s21");
    for (int it = 0; it < kIters; it++) {
'''
TAIL = r'''    }
    float s = a + b;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i] + (float)d[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int KIND>
void run(const char* name, int per_copy, float* d_out, int cus, double ghz) {
    printf("%-58s", name);
    for (int waves : {1, 2, 3, 4, 5, 8}) {
        const int blocks = cus * waves; // a 256-thread workgroup puts one wavefront on each SIMD of a CU
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        k_rate<KIND><<<blocks, 256>>>(d_out, 1.5f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_rate<KIND><<<blocks, 256>>>(d_out, 1.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf(" %6.2f", ms * 1e-3 * ghz * 1e9 / ((double)kIters * 16 * per_copy * waves));
    }
    printf("\n");
}
int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%d CUs at %.2f GHz; cycles of one SIMD per wave-instruction with 1 2 3 4 5 8 wavefronts per SIMD\n", cus, ghz);
    float* d_out;
    CHECK(hipMalloc(&d_out, 256));
    for (int i = 0; i < 300; i++) k_rate<0><<<cus * 8, 256>>>(d_out, 1.5f); // about a third of a second: lets the clocks settle
    CHECK(hipDeviceSynchronize());
'''


def main():
    src = [HEAD]
    calls = []
    for k, (label, pattern, count, clobber) in enumerate(KINDS):
        cl = f' : "{clobber}"' if clobber else ""
        asm = (f'asm volatile("{pattern}" : "+v"(v[i]), "+v"(a), "+v"(b), "+v"(v[(i + 5) & 15]), "+v"(v[(i + 9) & 15]), '
               f'"+v"(p[i]), "+v"(d[i]), "+v"(v[0]) :{cl});')
        src.append(f"        if (KIND == {k}) {{\n#pragma unroll\n            for (int i = 0; i < 16; i++) {{ {asm} }}\n        }}\n")
        calls.append(f'    run<{k}>("{label}", {count}, d_out, cus, ghz);\n')
    src.append(TAIL)
    src.extend(calls)
    src.append("    return 0;\n}\n")
    path = os.path.join(HERE, "valu_rate_gen.hip")
    with open(path, "w") as f:
        f.write("".join(src))
    out = os.path.join(HERE, "valu_rate_bin")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-o", out, path]
    print(" ".join(cmd))
    rc = subprocess.call(cmd)
    os.unlink(path)
    sys.exit(rc)


if __name__ == "__main__":
    main()
