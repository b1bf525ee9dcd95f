/*
 * ubench_ops.hip -- per-opcode issue cost on gfx950 (companion of ubench_issue.hip): every kind
 * is a 4-instruction pattern repeated 8x inside a loop, 8 waves per SIMD, all CUs busy.
 * Prints shader cycles per wave-instruction per SIMD (2.4 GHz assumed) at W = 2, 4, 8.
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define R8(x) x x x x x x x x

extern __shared__ char lds[];

template <int K> struct Body;
/* %0-%3 float a0..a3, %4-%7 int i0..i3, %8-%11 u64 m0..m3, %12 float b, %13 float c */
#define KIND(id, str) \
	template <> struct Body<id> { \
		static __device__ __forceinline__ void run(float &a0, float &a1, float &a2, float &a3, int &i0, int &i1, int &i2, int &i3, \
				unsigned long long &m0, unsigned long long &m1, unsigned long long &m2, unsigned long long &m3, float b, float c) { \
			asm volatile(R8(str) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), \
					"+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(b), "v"(c) : "vcc", "scc"); \
		} \
		static const char *name() { return str; } \
	};

#define V4(op) op " %0, %0, %12\n" op " %1, %1, %12\n" op " %2, %2, %12\n" op " %3, %3, %12\n"
#define I4(op) op " %4, %4, %5\n" op " %5, %5, %6\n" op " %6, %6, %7\n" op " %7, %7, %4\n"

KIND(1, V4("v_add_f32"))
KIND(2, V4("v_max_f32"))
KIND(3, V4("v_min_f32"))
KIND(4, V4("v_mul_f32"))
KIND(5, "v_fma_f32 %0, %0, %12, %13\n v_fma_f32 %1, %1, %12, %13\n v_fma_f32 %2, %2, %12, %13\n v_fma_f32 %3, %3, %12, %13\n")
KIND(6, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n")
KIND(7, "v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7\n")
KIND(8, I4("v_add_u32"))
KIND(9, I4("v_and_b32"))
KIND(10, I4("v_lshlrev_b32"))
KIND(11, "v_cndmask_b32 %0, %0, %12, vcc\n v_cndmask_b32 %1, %1, %12, vcc\n v_cndmask_b32 %2, %2, %12, vcc\n v_cndmask_b32 %3, %3, %12, vcc\n")
KIND(12, "v_cndmask_b32 %0, %0, %12, %8\n v_cndmask_b32 %1, %1, %12, %9\n v_cndmask_b32 %2, %2, %12, %10\n v_cndmask_b32 %3, %3, %12, %11\n")
KIND(13, "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0\n")
KIND(14, "v_cmp_gt_f32 %8, %0, %1\n v_cmp_gt_f32 %9, %1, %2\n v_cmp_gt_f32 %10, %2, %3\n v_cmp_gt_f32 %11, %3, %0\n")
KIND(15, "v_max3_f32 %0, %0, %1, %12\n v_max3_f32 %1, %1, %2, %12\n v_max3_f32 %2, %2, %3, %12\n v_max3_f32 %3, %3, %0, %12\n")
KIND(16, V4("v_sub_f32"))
KIND(17, "v_bfe_u32 %4, %5, 8, 8\n v_bfe_u32 %5, %6, 8, 8\n v_bfe_u32 %6, %7, 8, 8\n v_bfe_u32 %7, %4, 8, 8\n")
KIND(18, "v_cmp_eq_u32_sdwa vcc, %4, %5 src0_sel:BYTE_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %5, %6 src0_sel:BYTE_1 src1_sel:DWORD\n"
		"v_cmp_eq_u32_sdwa vcc, %6, %7 src0_sel:BYTE_1 src1_sel:DWORD\n v_cmp_eq_u32_sdwa vcc, %7, %4 src0_sel:BYTE_1 src1_sel:DWORD\n")
KIND(19, "v_add_f32_e64 %0, %0, %12\n v_add_f32_e64 %1, %1, %12\n v_add_f32_e64 %2, %2, %12\n v_add_f32_e64 %3, %3, %12\n")
KIND(20, "v_max_f32 %0, %0, %12\n v_max_f32 %0, %0, %13\n v_max_f32 %0, %0, %12\n v_max_f32 %0, %0, %13\n")
KIND(21, "v_add_f32 %0, %0, %12\n v_max_f32 %1, %1, %12\n v_add_f32 %2, %2, %12\n v_max_f32 %3, %3, %12\n")
KIND(22, "v_add_f32 %0, %0, %12\n v_cndmask_b32 %1, %1, %12, %8\n v_add_f32 %2, %2, %12\n v_cndmask_b32 %3, %3, %12, %9\n")
KIND(23, "v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_add_f32 %2, %2, %12\n s_and_b64 %8, %8, %9\n")
KIND(24, "s_lshl_b64 %8, %8, 1\n s_lshl_b64 %9, %9, 1\n s_not_b64 %10, %10\n s_mov_b64 %11, %8\n")
KIND(25, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
KIND(26, "v_nop\n v_nop\n v_nop\n v_nop\n")
KIND(27, "v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_addc_co_u32 %5, vcc, %5, %5, vcc\n v_addc_co_u32 %6, vcc, %6, %6, vcc\n v_addc_co_u32 %7, vcc, %7, %7, vcc\n")
KIND(28, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
		"v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KIND(29, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
KIND(30, "v_med3_f32 %0, %0, %1, %12\n v_med3_f32 %1, %1, %2, %12\n v_med3_f32 %2, %2, %3, %12\n v_med3_f32 %3, %3, %0, %12\n")
KIND(31, I4("v_max_i32"))
KIND(32, I4("v_max_u32"))
KIND(33, I4("v_or_b32"))
KIND(34, "v_lshl_or_b32 %4, %4, 1, %5\n v_lshl_or_b32 %5, %5, 1, %6\n v_lshl_or_b32 %6, %6, 1, %7\n v_lshl_or_b32 %7, %7, 1, %4\n")
KIND(35, "v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %4\n v_add3_u32 %7, %7, %4, %5\n")
KIND(36, "v_lshl_add_u32 %4, %4, 1, %5\n v_lshl_add_u32 %5, %5, 1, %6\n v_lshl_add_u32 %6, %6, 1, %7\n v_lshl_add_u32 %7, %7, 1, %4\n")
KIND(37, "v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_add_f32 %2, %2, %12\n v_cmp_gt_f32 %8, %3, %0\n")
KIND(38, "v_add_f32 %0, %0, %12\n s_and_b64 %8, %8, %9\n v_max_f32 %1, %1, %12\n s_or_b64 %9, %9, %10\n")
KIND(39, "v_cmp_gt_f32 %8, %0, %1\n s_and_b64 %9, %9, %10\n v_cndmask_b32 %2, %2, %12, %11\n s_or_b64 %10, %10, %11\n")
KIND(40, "v_cmp_lt_i32 vcc, %4, %5\n v_cmp_lt_i32 vcc, %5, %6\n v_cmp_lt_i32 vcc, %6, %7\n v_cmp_lt_i32 vcc, %7, %4\n")
KIND(41, "v_sub_u32 %4, %4, %5\n v_subrev_u32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_lshrrev_b32 %7, 1, %4\n")
KIND(42, "v_max_f32 %0, %0, %12\n v_max_f32 %1, %1, %12\n v_cmp_gt_f32 vcc, %2, %3\n v_cndmask_b32 %3, %3, %12, vcc\n")
KIND(43, "v_mul_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_mul_f32 %2, %2, %12\n v_add_f32 %3, %3, %12\n")
KIND(44, "v_min_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_min_f32 %2, %2, %12\n v_add_f32 %3, %3, %12\n")
KIND(45, "v_fmac_f32 %0, %1, %12\n v_fmac_f32 %1, %2, %12\n v_fmac_f32 %2, %3, %12\n v_fmac_f32 %3, %0, %12\n")
KIND(46, "v_add_f16 %0, %0, %12\n v_add_f16 %1, %1, %12\n v_add_f16 %2, %2, %12\n v_add_f16 %3, %3, %12\n")
KIND(47, "v_pk_add_f16 %0, %0, %12\n v_pk_add_f16 %1, %1, %12\n v_pk_max_f16 %2, %2, %12\n v_pk_max_f16 %3, %3, %12\n")
KIND(48, "v_pk_add_i16 %4, %4, %5\n v_pk_max_i16 %5, %5, %6\n v_pk_add_i16 %6, %6, %7\n v_pk_max_i16 %7, %7, %4\n")

template <int K>
__global__ void __launch_bounds__(256) kern(float *out, int iters, int never) {
	float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
	float b = 1.0f + (float) never, c = 0.5f + (float) never;
	unsigned long long m0 = 0x5555555555555555ull + (unsigned) never, m1 = 0x3333333333333333ull + (unsigned) never;
	unsigned long long m2 = m0 ^ 0xffull, m3 = m1 ^ 0xff00ull;
	int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
	for (int it = 0; it < iters; ++it) Body<K>::run(a0, a1, a2, a3, i0, i1, i2, i3, m0, m1, m2, m3, b, c);
	float s = a0 + a1 + a2 + a3 + (float) (i0 + i1 + i2 + i3) + (float) (unsigned) (m0 ^ m1 ^ m2 ^ m3);
	if (never) lds[threadIdx.x] = (char) s;
	if (s == 123.456f || never) out[blockIdx.x * blockDim.x + threadIdx.x] = s + (never ? lds[0] : 0);
}

template <int K>
static void run(float *out, double body = 32.0) {
	const int iters = 4096;
	char nm[64];
	const char *full = Body<K>::name();
	int n = 0;
	for (const char *p = full; *p && n < 58; ++p) nm[n++] = (*p == '\n') ? ';' : *p;
	nm[n] = 0;
	printf("%2d %-60s", K, nm);
	for (int W : {2, 4, 8}) {
		const int lds_bytes = (int) (160 * 1024 / W) - 1024;
		CHECK(hipFuncSetAttribute((const void *) kern<K>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		const int grid = 256 * W * 4;
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		hipLaunchKernelGGL(kern<K>, dim3(grid), dim3(256), lds_bytes, 0, out, 64, 0);
		CHECK(hipDeviceSynchronize());
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(kern<K>, dim3(grid), dim3(256), lds_bytes, 0, out, iters, 0);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		const double instr = 4.0 * W * iters * body;
		printf("  W=%d %5.2f", W, ms * 1e-3 * 2.4e9 / instr);
	}
	printf("\n");
}

template <int K, int KEND>
struct All { static void go(float *out) { run<K>(out); All<K + 1, KEND>::go(out); } };
template <int KEND>
struct All<KEND, KEND> { static void go(float *) {} };

int main() {
	float *out;
	CHECK(hipMalloc(&out, 256 * 8 * 4 * 256 * sizeof(float)));
	/* warm the clocks */
	for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern<1>, dim3(8192), dim3(256), 1024, 0, out, 4096, 0);
	CHECK(hipDeviceSynchronize());
	printf("cycles per wave-instruction per SIMD (2.4 GHz assumed)\n");
	All<1, 49>::go(out);
	return 0;
}
