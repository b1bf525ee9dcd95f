// tools/copy_probe.cpp -- which engine carries hipMemcpyAsync between pinned host memory and HBM here
// (SDMA or the __amd_rocclr_copyBuffer blit kernel), and at what rate, idle and beside a busy kernel queue.
//   hipcc -O2 --offload-arch=gfx950 tools/copy_probe.cpp -o tools/bin/copy_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(float *p, int n) {
	float v = p[threadIdx.x];
	for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f;
	p[threadIdx.x] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
	const size_t MB = 1 << 20;
	int lo, hi;
	CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	hipStream_t s_io, s_k;
	CK(hipStreamCreateWithPriority(&s_io, hipStreamNonBlocking, hi));
	CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
	void *h, *d; float *dk;
	CK(hipHostMalloc(&h, 512 * MB, hipHostMallocDefault));
	CK(hipMalloc(&d, 512 * MB));
	CK(hipMalloc((void **) &dk, 4096));
	memset(h, 1, 512 * MB);
	for (int busy = 0; busy < 2; ++busy)
		for (size_t sz : {size_t(1) * MB, 32 * MB, 320 * MB})
			for (int dir = 0; dir < 2; ++dir) {
				if (busy) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s_k, dk, 200000);
				CK(hipStreamSynchronize(s_io));
				double t0 = now();
				for (int r = 0; r < 4; ++r) {
					if (dir == 0) CK(hipMemcpyAsync(d, h, sz, hipMemcpyHostToDevice, s_io));
					else CK(hipMemcpyAsync(h, d, sz, hipMemcpyDeviceToHost, s_io));
				}
				CK(hipStreamSynchronize(s_io));
				double dt = (now() - t0) / 4;
				printf("%s %4zu MB %s: %.3f ms  %.1f GB/s\n", busy ? "busy" : "idle", sz / MB, dir ? "D2H" : "H2D", dt * 1e3, sz / dt / 1e9);
				CK(hipDeviceSynchronize());
			}
	// does a kernel on the SAME stream change the engine?  (look for __amd_rocclr_copyBuffer in a kernel trace)
	for (int mixed = 0; mixed < 2; ++mixed) {
		CK(hipDeviceSynchronize());
		double t0 = now();
		for (int r = 0; r < 4; ++r) {
			if (mixed) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s_io, dk, 10);
			CK(hipMemcpyAsync(d, h, 64 * MB, hipMemcpyHostToDevice, s_io));
			CK(hipMemcpyAsync((char *) d + 64 * MB, (char *) h + 64 * MB, 256 * MB, hipMemcpyHostToDevice, s_io));
		}
		CK(hipStreamSynchronize(s_io));
		printf("%s stream, 4 x (64 + 256 MB) H2D: %.3f ms\n", mixed ? "kernel+copy" : "copy-only", (now() - t0) * 1e3);
	}
	// after a device-wide synchronize that followed kernels on another stream
	for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(spin, dim3(4096), dim3(256), 0, s_k, dk, 20000);
	CK(hipDeviceSynchronize());
	double t1 = now();
	CK(hipMemcpyAsync(d, h, 256 * MB, hipMemcpyHostToDevice, s_io));
	CK(hipStreamSynchronize(s_io));
	printf("after device sync, 256 MB H2D: %.3f ms\n", (now() - t1) * 1e3);
	return 0;
}
