/*
 * ubench_gather.hip -- what "HBM random access" is worth on an MI355X.
 *
 * The candidate search at genome scale (cvx_search.hip over a 1 GB k-mer table: one 8-byte index record per k-mer and
 * orientation, then a short run of 4-byte locations per row) is bound by reads of sectors that nothing else in the launch
 * touches.  This measures that ceiling by itself: every lane reads `bytes` (8 / 16 / 64) at a pseudo-random sector-aligned
 * address of a buffer far larger than the 256 MB of last-level cache, with 1 / 4 / 16 independent loads in flight per lane,
 * at 1-8 waves per SIMD.  Output: G accesses / s and the GB/s of 64-byte sectors they pull (one sector per access whatever
 * the bytes used); bench.py's index_stage_device.candidate_search_big quotes the best row as `random_sector_peak` and the
 * search kernel's own sector rate against it.
 *
 *   hipcc -O2 --offload-arch=gfx950 tools/ubench_gather.hip -o tools/bin/ubench_gather && tools/bin/ubench_gather [GiB]
 * One line per configuration; the last line is machine-readable:  RANDOM_SECTOR_PEAK <G accesses/s> <GB/s of 64 B sectors>
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {      /* splitmix64: the address stream */
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

template <int INFLIGHT, int BYTES>
__global__ void __launch_bounds__(256) gather(const uint8_t *buf, uint64_t n_sectors, int iters, uint32_t *out) {
	uint64_t s = mix((uint64_t) blockIdx.x * 256u + threadIdx.x);
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		uint64_t idx[INFLIGHT];
#pragma unroll
		for (int k = 0; k < INFLIGHT; ++k) { s = mix(s); idx[k] = (s % n_sectors) * 64ull; }
#pragma unroll
		for (int k = 0; k < INFLIGHT; ++k) {
			if (BYTES == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(buf + idx[k]); acc += v.x ^ v.y; }
			else if (BYTES == 16) { const uint4 v = *reinterpret_cast<const uint4 *>(buf + idx[k]); acc += v.x ^ v.w; }
			else {
#pragma unroll
				for (int q = 0; q < 4; ++q) { const uint4 v = *reinterpret_cast<const uint4 *>(buf + idx[k] + 16 * q); acc += v.x ^ v.w; }
			}
		}
	}
	out[blockIdx.x * 256u + threadIdx.x] = acc;
}

static double best_acc = 0.0;

template <int INFLIGHT, int BYTES>
static void run(const uint8_t *buf, uint64_t n_sectors, uint32_t *out) {
	printf("%2d B per access, %2d loads in flight per lane:", BYTES, INFLIGHT);
	for (int W : {1, 2, 4, 8}) {
		const int grid = 256 * W;               /* 256 CUs x W workgroups of 4 waves = W waves per SIMD */
		const int iters = 2048 / INFLIGHT;
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		hipLaunchKernelGGL((gather<INFLIGHT, BYTES>), dim3(grid), dim3(256), 0, 0, buf, n_sectors, 8, out);
		CHECK(hipDeviceSynchronize());
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((gather<INFLIGHT, BYTES>), dim3(grid), dim3(256), 0, 0, buf, n_sectors, iters, out);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		const double acc = (double) grid * 256.0 * iters * INFLIGHT / (ms * 1e-3);
		if (acc > best_acc) best_acc = acc;
		printf("  W=%d %6.2f G/s (%5.0f GB/s of sectors)", W, acc * 1e-9, acc * 64.0 * 1e-9);
		CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
	}
	printf("\n");
}

int main(int argc, char **argv) {
	const double gib = argc > 1 ? atof(argv[1]) : 4.0;
	const uint64_t bytes = (uint64_t) (gib * 1024.0 * 1024.0 * 1024.0) / 64 * 64;
	uint8_t *buf;
	uint32_t *out;
	CHECK(hipMalloc(&buf, bytes));
	CHECK(hipMemset(buf, 1, bytes));
	CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(uint32_t)));
	printf("random 64-byte-sector reads over %.1f GiB, W = waves per SIMD\n", gib);
	run<1, 8>(buf, bytes / 64, out);
	run<4, 8>(buf, bytes / 64, out);
	run<16, 8>(buf, bytes / 64, out);
	run<4, 16>(buf, bytes / 64, out);
	run<16, 16>(buf, bytes / 64, out);
	run<4, 64>(buf, bytes / 64, out);
	run<16, 64>(buf, bytes / 64, out);
	printf("RANDOM_SECTOR_PEAK %.3f %.1f\n", best_acc * 1e-9, best_acc * 64.0 * 1e-9);
	return 0;
}
