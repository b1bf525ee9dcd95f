// Dev tool (GPU box): what a host thread costs while it waits for the device.  A kernel that runs ~ms_target, then one of the
// runtime's waits; prints wall clock and the waiting thread's own CPU time (CLOCK_THREAD_CPUTIME_ID).
//   hipcc --offload-arch=gfx950 -O2 -o wait_probe tools/wait_probe.hip && ./wait_probe [ms]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <unistd.h>

__global__ void spin_kernel(unsigned long long ticks, unsigned long long *out) {
	unsigned long long t0 = wall_clock64();
	unsigned long long t = t0;
	while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(100); t = wall_clock64(); }
	if (threadIdx.x == 0) out[0] = t - t0;
}

static double now_wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static double now_cpu() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
	const double ms = argc > 1 ? atof(argv[1]) : 50.0;
	const char *flags = getenv("PROBE_DEVICE_FLAGS");
	if (flags && atoi(flags) == 1) CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
	if (flags && atoi(flags) == 2) CK(hipSetDeviceFlags(hipDeviceScheduleYield));
	CK(hipSetDevice(0));
	hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	unsigned long long *d; CK(hipMalloc(&d, 8));
	int rate_khz = 0; CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
	const unsigned long long ticks = (unsigned long long) (ms * 1e-3 * rate_khz * 1e3);
	hipEvent_t plain, blocking; CK(hipEventCreateWithFlags(&plain, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&blocking, hipEventBlockingSync | hipEventDisableTiming));
	printf("wall clock rate %d kHz, kernel of %.0f ms, device flags %s\n", rate_khz, ms, flags ? flags : "default");
	for (int mode = 0; mode < 5; ++mode) {
		double cw = 0, cc = 0;
		for (int rep = 0; rep < 4; ++rep) {
			hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, ticks, d);
			if (mode == 1) CK(hipEventRecord(plain, st));
			if (mode == 2 || mode == 4) CK(hipEventRecord(blocking, st));
			const double w0 = now_wall(), c0 = now_cpu();
			if (mode == 0) CK(hipStreamSynchronize(st));
			if (mode == 1) CK(hipEventSynchronize(plain));
			if (mode == 2) CK(hipEventSynchronize(blocking));
			if (mode == 3) { while (hipStreamQuery(st) == hipErrorNotReady) usleep(200); }
			if (mode == 4) { while (hipEventQuery(blocking) == hipErrorNotReady) usleep(200); }
			if (rep) { cw += now_wall() - w0; cc += now_cpu() - c0; }
		}
		const char *names[5] = {"hipStreamSynchronize", "hipEventSynchronize (plain event)", "hipEventSynchronize (hipEventBlockingSync)", "hipStreamQuery + usleep(200)", "hipEventQuery + usleep(200)"};
		printf("%-44s wall %7.2f ms  thread cpu %7.2f ms per wait\n", names[mode], cw / 3 * 1e3, cc / 3 * 1e3);
	}
	return 0;
}
