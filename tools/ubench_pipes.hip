/*
 * ubench_pipes.hip -- which VALU instructions of the fill kernel can run beside which (gfx950).
 *
 * Round 2 left a discrepancy: alternating a full-rate op (v_add_f32 ...) with a half-rate one (v_max_f32,
 * v_cmp, v_cndmask, v_addc, DPP ...) costs the half-rate op's time alone in profiles/r01_ubench_ops.txt,
 * while inside fill_ring_kernel the two classes add up.  This benchmark measures in SHADER CYCLES (s_memtime
 * around the loop of every wave; wall-clock rates under an assumed 2.4 GHz move with the power state), with
 *   - simple streams: F only, S only, alternating, 2:1 in several groupings, dependent chains;
 *   - heterogeneous waves: the two waves of a SIMD running an S-only and an F-only stream;
 *   - the fill kernel's own cell update (three slots, real operand forms, SALU mask logic in between) in the
 *     compiler's order, interleaved, list-scheduled to alternate the classes, and with each class removed
 *     (tools/gen_ubench_pipes.py writes tools/ubench_pipes_bodies.inc).
 * Output: cycles per loop body per SIMD (= per-wave cycles / waves per SIMD) and per VALU instruction.
 *
 *   hipcc -O2 --offload-arch=gfx950 tools/ubench_pipes.hip -o tools/bin/ubench_pipes
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define CLOB "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", \
	"v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", \
	"v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
	"v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", \
	"s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", \
	"s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "vcc", "scc"

template <int K> struct Body;
#define KIND(id, nm, body, ns, nf, nx) \
	template <> struct Body<id> { \
		static __device__ __forceinline__ void run() { asm volatile(body ::: CLOB); } \
		static const char *name() { return nm; } \
		static constexpr int nS = ns, nF = nf, nX = nx; \
	};
#include "ubench_pipes_bodies.inc"

/* KA != KB: 512-thread workgroups, waves 0-3 run KA, waves 4-7 (the second wave of each SIMD) run KB */
template <int KA, int KB>
__global__ void __launch_bounds__(512) kern(long long *cycles, int iters) {
	extern __shared__ char lds[];
	asm volatile("v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 0.15\n v_mov_b32 v11, -5.0\n v_mov_b32 v12, -1.0\n v_mov_b32 v13, -5.0\n" ::: "v8", "v9", "v10", "v11", "v12", "v13");
	const int wave = threadIdx.x >> 6;
	__syncthreads();
	const long long t0 = clock64();
	if (KA == KB || wave < 4) {
		for (int it = 0; it < iters; ++it) Body<KA>::run();
	} else {
		for (int it = 0; it < iters; ++it) Body<KB>::run();
	}
	const long long t1 = clock64();
	if ((threadIdx.x & 63) == 0) cycles[(size_t) blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
	if (iters < 0) lds[threadIdx.x] = 1;
}

static double g_ticks_per_ns = 0.0;

/* returns s_memtime ticks per body per wave (A half; *cb the B half when hetero) and the wall time */
template <int KA, int KB>
static double run(long long *d_cycles, int W, bool hetero, int iters, double *cb, double *wall_ms) {
	const int threads = hetero ? 512 : 256;
	const int wg_per_cu = hetero ? W / 2 : W;
	const int lds_bytes = (int) (160 * 1024 / wg_per_cu) - 1024;
	CHECK(hipFuncSetAttribute((const void *) kern<KA, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
	const int grid = 256 * wg_per_cu;
	const int waves = grid * threads / 64;
	hipLaunchKernelGGL((kern<KA, KB>), dim3(grid), dim3(threads), lds_bytes, 0, d_cycles, iters / 8 + 1);
	CHECK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL((kern<KA, KB>), dim3(grid), dim3(threads), lds_bytes, 0, d_cycles, iters);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	if (wall_ms) *wall_ms = ms;
	std::vector<long long> h((size_t) waves);
	CHECK(hipMemcpy(h.data(), d_cycles, (size_t) waves * sizeof(long long), hipMemcpyDeviceToHost));
	double sa = 0, sb = 0;
	long na = 0, nb = 0;
	const int wpb = threads / 64;
	for (int i = 0; i < waves; ++i) {
		if (hetero && (i % wpb) >= 4) { sb += (double) h[(size_t) i]; nb++; }
		else { sa += (double) h[(size_t) i]; na++; }
	}
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
	if (cb) *cb = nb ? sb / (double) nb / iters : 0.0;
	return sa / (double) na / iters;
}

template <int K>
static void one(long long *d) {
	const int n = Body<K>::nS + Body<K>::nF + Body<K>::nX;
	const int iters = n > 60 ? 1500 : 6000;
	printf("%3d %-52s S=%2d F=%2d SALU=%2d |", K, Body<K>::name(), Body<K>::nS, Body<K>::nF, Body<K>::nX);
	for (int W : {4, 6, 8}) {
		double wall = 0;
		const double c = run<K, K>(d, W, false, iters, nullptr, &wall);
		/* ticks per body per SIMD; wall-clock ns per body per SIMD */
		printf("  W=%d %7.1f tk %7.1f ns", W, c / W, wall * 1e6 / iters / W);
	}
	printf("\n");
}

template <int K, int KEND> struct All { static void go(long long *d) { one<K>(d); All<K + 1, KEND>::go(d); } };
template <int KEND> struct All<KEND, KEND> { static void go(long long *) {} };

int main() {
	long long *d;
	CHECK(hipMalloc(&d, (size_t) 256 * 8 * 8 * sizeof(long long)));
	for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kern<0, 0>), dim3(2048), dim3(256), 1024, 0, d, 20000);
	CHECK(hipDeviceSynchronize());
	{
		/* what a tick is: a long F-only run, ticks against wall time */
		double wall = 0;
		const double c = run<0, 0>(d, 8, false, 200000, nullptr, &wall);
		g_ticks_per_ns = c * 200000 / (wall * 1e6);
		printf("s_memtime: %.4f ticks per ns of wall time (long F-only run at W=8: %.1f ticks per 24-instruction body per wave)\n", g_ticks_per_ns, c);
	}
	printf("per loop body per SIMD: s_memtime ticks (tk) and wall-clock ns; W = waves per SIMD\n");
	All<0, kKinds>::go(d);
	printf("heterogeneous waves on one SIMD: A on waves 0-3, B on waves 4-7 of 512-thread groups; ticks per body per wave\n");
	for (int W : {2, 4, 8}) {
		double cb = 0;
		double ca = run<1, 0>(d, W, true, 6000, &cb, nullptr);
		printf("  W=%d  A=v_max x24: %7.1f  B=v_add x24: %7.1f", W, ca, cb);
		ca = run<1, 1>(d, W, false, 6000, nullptr, nullptr);
		const double cf = run<0, 0>(d, W, false, 6000, nullptr, nullptr);
		printf("   (homogeneous at the same W: v_max %7.1f, v_add %7.1f)\n", ca, cf);
	}
	return 0;
}
