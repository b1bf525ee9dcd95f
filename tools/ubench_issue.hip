/*
 * ubench_issue.hip -- instruction-issue microbenchmark for gfx950, used to price the forward
 * fill (DESIGN.md "what bounds the fill").  Each kernel runs a 32-instruction inline-asm body
 * in a loop; occupancy is pinned with dynamic LDS (W workgroups of 4 waves per CU = W waves
 * per SIMD).  Prints shader cycles per instruction per SIMD assuming 2.4 GHz.
 *
 *   hipcc -O2 --offload-arch=gfx950 tools/ubench_issue.hip -o /tmp/ubench_issue && /tmp/ubench_issue
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)

extern __shared__ char lds[];

template <int KIND>
__global__ void __launch_bounds__(256) kern(float *out, int iters, int never) {
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	float b = 1.0f + (float) never;
	unsigned long long m0 = 0x5555555555555555ull + (unsigned) never, m1 = 0x3333333333333333ull + (unsigned) never;
	unsigned long long m2 = m0 ^ 0xffull, m3 = m1 ^ 0xff00ull;
	int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
	for (int it = 0; it < iters; ++it) {
		if constexpr (KIND == 0) {       /* v_add_f32, 8 independent chains */
			asm volatile(R4("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
					"v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
		} else if constexpr (KIND == 1) { /* v_max_f32 */
			asm volatile(R4("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
					"v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
		} else if constexpr (KIND == 2) { /* v_cndmask_b32 with an SGPR-pair mask */
			asm volatile(R4("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
					"v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(m0));
		} else if constexpr (KIND == 3) { /* v_cmp_gt_f32 into SGPR pairs */
			asm volatile(R8("v_cmp_gt_f32 %0, %4, %5\n v_cmp_gt_f32 %1, %5, %6\n v_cmp_gt_f32 %2, %6, %7\n v_cmp_gt_f32 %3, %7, %4\n")
					: "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
		} else if constexpr (KIND == 4) { /* SALU only: 4 independent 64-bit chains */
			asm volatile(R8("s_and_b64 %0, %0, %1\n s_or_b64 %1, %1, %2\n s_xor_b64 %2, %2, %3\n s_andn2_b64 %3, %3, %0\n")
					: "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");
		} else if constexpr (KIND == 5) { /* 2 VALU : 1 SALU, independent */
			asm volatile(R8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n s_and_b64 %4, %4, %5\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n s_or_b64 %5, %5, %4\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+v"(a6), "+v"(a7) : "v"(b) : "scc");
		} else if constexpr (KIND == 6) { /* the fill's idiom: cmp -> s_and -> cndmask, dependent */
			asm volatile(R8("v_cmp_eq_f32 %4, %0, %1\n s_and_b64 %5, %4, %6\n v_cndmask_b32 %2, %2, %8, %5\n v_max_f32 %0, %0, %2\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(b) : "scc");
		} else if constexpr (KIND == 7) { /* v_addc_co_u32_e64 with SGPR carry in/out */
			asm volatile(R8("v_addc_co_u32_e64 %0, %4, %0, %0, %6\n v_addc_co_u32_e64 %1, %5, %1, %1, %7\n v_addc_co_u32_e64 %2, %4, %2, %2, %6\n v_addc_co_u32_e64 %3, %5, %3, %3, %7\n")
					: "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+s"(m0), "+s"(m1) : "s"(m2), "s"(m3));
		} else if constexpr (KIND == 8) { /* DPP move (wave_ror:1) */
			asm volatile(R8("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
					"v_mov_b32_dpp %2, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
		} else if constexpr (KIND == 9) { /* gap penalty: cvt, mul, add, min */
			asm volatile(R8("v_cvt_f32_i32 %0, %4\n v_mul_f32 %1, %0, %8\n v_add_f32 %2, %1, %8\n v_min_f32 %3, %2, %8\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(a6), "+v"(a7) : "v"(b));
		} else if constexpr (KIND == 10) { /* 1 VALU : 1 SALU independent */
			asm volatile(R16("v_add_f32 %0, %0, %8\n s_and_b64 %4, %4, %5\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+v"(a6), "+v"(a7) : "v"(b) : "scc");
		} else if constexpr (KIND == 11) { /* v_cmp writing VCC, cndmask reading VCC */
			asm volatile(R16("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %8, vcc\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
		} else if constexpr (KIND == 12) { /* v_cmpx-free masked move: s_mov exec, v_mov, restore */
			asm volatile(R8("s_mov_b64 exec, %4\n v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n s_mov_b64 exec, -1\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+v"(a6), "+v"(a7) : "v"(b));
		} else if constexpr (KIND == 13) { /* packed f32 add (2 lanes-worth per op) */
			asm volatile(R8("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n")
					: "+v"(*(double *) &a0), "+v"(*(double *) &a2) : "v"(*(double *) &a4));
		} else if constexpr (KIND == 14) { /* 3-input max */
			asm volatile(R8("v_max3_f32 %0, %0, %1, %8\n v_max3_f32 %1, %1, %2, %8\n v_max3_f32 %2, %2, %3, %8\n v_max3_f32 %3, %3, %0, %8\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+v"(a6), "+v"(a7) : "v"(b));
		} else if constexpr (KIND == 16) { /* one dependent VALU chain */
			asm volatile(R32("v_add_f32 %0, %0, %8\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+v"(a6), "+v"(a7) : "v"(b));
		} else if constexpr (KIND == 17) { /* one dependent SALU chain */
			asm volatile(R16("s_and_b64 %4, %4, %5\n s_or_b64 %4, %4, %6\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");
		} else if constexpr (KIND == 18) { /* three independent copies of the fill idiom (M = 3 slots) */
			asm volatile(R4("v_cmp_eq_f32 %4, %0, %3\n v_cmp_eq_f32 %5, %1, %3\n v_cmp_eq_f32 %6, %2, %3\n"
					"s_and_b64 %4, %4, %7\n s_and_b64 %5, %5, %7\n s_and_b64 %6, %6, %7\n"
					"v_cndmask_b32 %0, %0, %8, %4\n v_cndmask_b32 %1, %1, %8, %5\n")
					: "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(b) : "scc");
		} else if constexpr (KIND == 15) { /* v_cmp_class-like e64 compare into SGPR then s_cselect-free bcnt */
			asm volatile(R8("v_cmp_lt_u32 %4, %0, %1\n v_cmp_lt_u32 %5, %1, %2\n v_cmp_lt_u32 %6, %2, %3\n v_cmp_lt_u32 %7, %3, %0\n")
					: "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3));
		}
	}
	float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float) (i0 + i1 + i2 + i3) + (float) (unsigned) (m0 ^ m1 ^ m2 ^ m3);
	if (never) lds[threadIdx.x] = (char) s;
	if (s == 123.456f || never) out[blockIdx.x * blockDim.x + threadIdx.x] = s + (never ? lds[0] : 0);
}

template <int KIND>
static void run(const char *name, float *out, double body = 32.0) {
	const int iters = 4096;
	printf("%-44s", name);
	for (int W : {1, 2, 4, 6, 8}) {
		const int lds_bytes = (int) (160 * 1024 / W) - 1024;
		CHECK(hipFuncSetAttribute((const void *) kern<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		const int grid = 256 * W * 4;
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		hipLaunchKernelGGL(kern<KIND>, dim3(grid), dim3(256), lds_bytes, 0, out, 64, 0);
		CHECK(hipDeviceSynchronize());
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(kern<KIND>, dim3(grid), dim3(256), lds_bytes, 0, out, iters, 0);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		/* per SIMD: 4 rounds x W waves x iters x 32 instructions */
		const double instr = 4.0 * W * iters * body;
		const double cyc = ms * 1e-3 * 2.4e9;
		printf("  W=%d %5.2f", W, cyc / instr);
	}
	printf("\n");
}

int main() {
	float *out;
	CHECK(hipMalloc(&out, 256 * 8 * 4 * 256 * sizeof(float)));
	printf("cycles per wave-instruction per SIMD (2.4 GHz assumed), W = waves per SIMD\n");
	run<0>("v_add_f32 x8 chains", out);
	run<1>("v_max_f32 x8 chains", out);
	run<14>("v_max3_f32", out);
	run<13>("v_pk_add_f32", out);
	run<2>("v_cndmask_b32 (sgpr mask)", out);
	run<3>("v_cmp_gt_f32 -> sgpr pair", out);
	run<15>("v_cmp_lt_u32 -> sgpr pair", out);
	run<11>("v_cmp->vcc, v_cndmask<-vcc (dependent)", out);
	run<4>("s_and/or/xor/andn2_b64", out);
	run<5>("2 VALU : 1 SALU independent", out, 48.0);
	run<10>("1 VALU : 1 SALU independent", out);
	run<6>("v_cmp->s_and->v_cndmask->v_max dependent", out);
	run<7>("v_addc_co_u32_e64 (sgpr carry)", out);
	run<8>("v_mov_b32_dpp wave_ror:1", out);
	run<9>("cvt,mul,add,min chain", out);
	run<12>("s_mov exec; 2 v_mov; s_mov exec", out);
	run<16>("dependent v_add_f32 chain", out);
	run<17>("dependent s_and/s_or chain", out);
	run<18>("3 slots x (v_cmp, s_and, v_cndmask) grouped", out);
	return 0;
}
