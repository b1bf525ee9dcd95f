/*
 * ubench_mix.hip -- how the two VALU classes of gfx950 (full-rate: f32 add/mul/fma, mov, logic,
 * add_u32; half-rate: everything else the fill kernel uses) share a SIMD.  ubench_ops.hip rows
 * 21/22/44 show that ONE wave alternating a half-rate with a full-rate op issues at ~2.2 cycles
 * per instruction (the pair overlaps); the fill kernel behaves as if the classes added up.  This
 * benchmark separates the two possible reasons:
 *   xw   : half of the waves of every SIMD run only half-rate ops, the other half only full-rate
 *          ops -- does the overlap work ACROSS waves?
 *   order: one wave, same multiset of independent ops, clustered (16 B then 16 A) vs alternating
 *   dep  : the alternating stream as one dependent chain (latency bound?) at 6 waves/SIMD
 *   cell : a v_cmp -> s_and -> v_cndmask -> v_add chain per "slot", 3 slots, slot-major vs interleaved
 * 512-thread blocks (8 waves: waves w and w+4 share a SIMD), 3 blocks per CU = 6 waves per SIMD.
 * Prints shader cycles per wave-instruction per SIMD (2.4 GHz assumed).
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define R4(x) x x x x
#define R8(x) x x x x x x x x

extern __shared__ char lds[];

#define REGS float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
	float b = 1.0f + (float) never; \
	unsigned long long m0 = 0x5555555555555555ull + (unsigned) never, m1 = 0x3333333333333333ull + (unsigned) never, m2 = m0 ^ 0xff, m3 = m1 ^ 0xff00;
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : "v"(b) : "vcc", "scc"
#define FIN float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float) (unsigned) (m0 ^ m1 ^ m2 ^ m3); \
	if (never) lds[threadIdx.x] = (char) s; \
	if (s == 123.456f || never) out[blockIdx.x * blockDim.x + threadIdx.x] = s + (never ? lds[0] : 0);

/* 8 independent ops on a0..a7 */
#define S8 "v_max_f32 %0, %0, %12\n v_max_f32 %1, %1, %12\n v_max_f32 %2, %2, %12\n v_max_f32 %3, %3, %12\n" \
           "v_max_f32 %4, %4, %12\n v_max_f32 %5, %5, %12\n v_max_f32 %6, %6, %12\n v_max_f32 %7, %7, %12\n"
#define F8 "v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_add_f32 %2, %2, %12\n v_add_f32 %3, %3, %12\n" \
           "v_add_f32 %4, %4, %12\n v_add_f32 %5, %5, %12\n v_add_f32 %6, %6, %12\n v_add_f32 %7, %7, %12\n"
#define ALT8 "v_max_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_max_f32 %2, %2, %12\n v_add_f32 %3, %3, %12\n" \
             "v_max_f32 %4, %4, %12\n v_add_f32 %5, %5, %12\n v_max_f32 %6, %6, %12\n v_add_f32 %7, %7, %12\n"
/* the alternating stream as ONE dependent chain */
#define DEP8 "v_max_f32 %0, %0, %12\n v_add_f32 %0, %0, %12\n v_max_f32 %0, %0, %12\n v_add_f32 %0, %0, %12\n" \
             "v_max_f32 %0, %0, %12\n v_add_f32 %0, %0, %12\n v_max_f32 %0, %0, %12\n v_add_f32 %0, %0, %12\n"
/* two dependent chains interleaved */
#define DEP2 "v_max_f32 %0, %0, %12\n v_max_f32 %1, %1, %12\n v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n" \
             "v_max_f32 %0, %0, %12\n v_max_f32 %1, %1, %12\n v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n"
/* three "slots", each cmp -> s_and -> cndmask -> add -> max; slot-major */
#define SLOT(a, m) "v_cmp_gt_f32 " m ", " a ", %12\n s_and_b64 " m ", " m ", %11\n v_cndmask_b32 " a ", " a ", %12, " m "\n v_add_f32 " a ", " a ", %12\n v_max_f32 " a ", " a ", %12\n"
#define CELL_SM SLOT("%0", "%8") SLOT("%1", "%9") SLOT("%2", "%10")
#define CELL_IL "v_cmp_gt_f32 %8, %0, %12\n v_cmp_gt_f32 %9, %1, %12\n v_cmp_gt_f32 %10, %2, %12\n" \
                "s_and_b64 %8, %8, %11\n s_and_b64 %9, %9, %11\n s_and_b64 %10, %10, %11\n" \
                "v_cndmask_b32 %0, %0, %12, %8\n v_cndmask_b32 %1, %1, %12, %9\n v_cndmask_b32 %2, %2, %12, %10\n" \
                "v_add_f32 %0, %0, %12\n v_max_f32 %1, %1, %12\n v_add_f32 %2, %2, %12\n" \
                "v_max_f32 %0, %0, %12\n v_add_f32 %1, %1, %12\n v_max_f32 %2, %2, %12\n"
/* skewed: every half-rate op is followed by an independent full-rate or scalar op */
#define CELL_SK "v_cmp_gt_f32 %8, %0, %12\n v_add_f32 %1, %1, %12\n v_cmp_gt_f32 %10, %2, %12\n s_and_b64 %8, %8, %11\n" \
                "v_max_f32 %1, %1, %12\n s_and_b64 %10, %10, %11\n v_cndmask_b32 %0, %0, %12, %8\n v_add_f32 %3, %3, %12\n" \
                "v_cndmask_b32 %2, %2, %12, %10\n v_add_f32 %0, %0, %12\n v_cmp_gt_f32 %9, %1, %12\n v_add_f32 %2, %2, %12\n" \
                "v_max_f32 %0, %0, %12\n s_and_b64 %9, %9, %11\n v_max_f32 %2, %2, %12\n"

enum { K_SLOW, K_FAST, K_XW, K_CLUST, K_ALT, K_DEP, K_DEP2, K_CELL_SM, K_CELL_IL, K_CELL_SK, K_N };
static const char *kNames[K_N] = {"all waves half-rate (v_max x8 indep)", "all waves full-rate (v_add x8 indep)",
	"xw: waves 0-3 half-rate, waves 4-7 full-rate", "order: 16 half then 16 full (indep)", "order: alternating half/full (indep)",
	"dep: alternating, one dependent chain", "dep: alternating, two chains", "cell x3 slot-major", "cell x3 phase-interleaved", "cell x3 skewed"};
static const double kInstr[K_N] = {32, 32, 32, 32, 32, 32, 32, 30, 30, 30};

template <int K>
__global__ void __launch_bounds__(512) kern(float *out, int iters, int never) {
	REGS
	const int w = threadIdx.x >> 6;
	if (K == K_XW && w >= 4) {
		for (int it = 0; it < iters; ++it) asm volatile(R4(F8) OPS);
	} else {
		for (int it = 0; it < iters; ++it) {
			if (K == K_SLOW || K == K_XW) asm volatile(R4(S8) OPS);
			if (K == K_FAST) asm volatile(R4(F8) OPS);
			if (K == K_CLUST) asm volatile(S8 S8 F8 F8 OPS);
			if (K == K_ALT) asm volatile(R4(ALT8) OPS);
			if (K == K_DEP) asm volatile(R4(DEP8) OPS);
			if (K == K_DEP2) asm volatile(R4(DEP2) OPS);
			if (K == K_CELL_SM) asm volatile(CELL_SM CELL_SM OPS);
			if (K == K_CELL_IL) asm volatile(CELL_IL CELL_IL OPS);
			if (K == K_CELL_SK) asm volatile(CELL_SK CELL_SK OPS);
		}
	}
	FIN
}

template <int K>
static void run(float *out) {
	const int iters = 4096;
	printf("%-48s", kNames[K]);
	for (int B : {1, 2, 3, 4}) {          /* blocks per CU -> 2, 4, 6, 8 waves per SIMD */
		const int lds_bytes = (int) (160 * 1024 / B) - 1024;
		CHECK(hipFuncSetAttribute((const void *) kern<K>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
		const int grid = 256 * B * 4;
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		hipLaunchKernelGGL(kern<K>, dim3(grid), dim3(512), lds_bytes, 0, out, 64, 0);
		CHECK(hipDeviceSynchronize());
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(kern<K>, dim3(grid), dim3(512), lds_bytes, 0, out, iters, 0);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		/* wave-instructions per SIMD: 4 rounds * B blocks * 2 waves per SIMD per block */
		const double instr = 4.0 * B * 2.0 * iters * kInstr[K];
		printf("  W=%d %5.2f", 2 * B, ms * 1e-3 * 2.4e9 / instr);
	}
	printf("\n");
}

template <int K> struct All { static void go(float *out) { run<K>(out); All<K + 1>::go(out); } };
template <> struct All<K_N> { static void go(float *) {} };

int main() {
	float *out;
	CHECK(hipMalloc(&out, (size_t) 256 * 8 * 4 * 512 * sizeof(float)));
	for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern<K_FAST>, dim3(4096), dim3(512), 1024, 0, out, 4096, 0);
	CHECK(hipDeviceSynchronize());
	printf("cycles per wave-instruction per SIMD (2.4 GHz assumed), W = waves per SIMD\n");
	All<0>::go(out);
	return 0;
}
