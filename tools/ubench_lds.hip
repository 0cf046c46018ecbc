/*
 * ubench_lds.hip -- LDS instruction cost on gfx950 for the access shapes the lz4 expand
 * kernel's window copies can be built from (measurement tool; not product).
 *
 * For every (instruction, address alignment, active lanes, waves per CU) it reports the
 * shader cycles one CU needs per wave-instruction when `waves` waves issue the instruction
 * back to back (8 per s_waitcnt) at per-lane pseudo-random addresses inside a 64 KiB window:
 * elapsed cycles of the slowest wave / (instructions per wave * waves on the CU).
 *
 * Build: hipcc -O3 --offload-arch=gfx950 -o ubench_lds ubench_lds.hip ; run: ./ubench_lds
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define ITERS 256	/* groups of 8 instructions per wave */

enum {
	OP_RD_B32, OP_RD_B64, OP_RD2_B32, OP_RD_B128, OP_RD_U8, OP_RD_U16,
	OP_WR_B32, OP_WR_B64, OP_WR2_B32, OP_WR_B8, OP_WR_B16, OP_MSKOR_B32, OP_MSKOR_B64, OP_OR_B32, OP_WR_B128,
	OP_COUNT
};
static const char *op_name[OP_COUNT] = {
	"ds_read_b32", "ds_read_b64", "ds_read2_b32", "ds_read_b128", "ds_read_u8", "ds_read_u16",
	"ds_write_b32", "ds_write_b64", "ds_write2_b32", "ds_write_b8", "ds_write_b16", "ds_mskor_b32", "ds_mskor_b64", "ds_or_b32", "ds_write_b128"
};

template <int OP>
__device__ __forceinline__ void group8(const uint32_t (&a)[8], uint32_t &sink)
{
	uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
	uint64_t q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0;
	const uint64_t d64 = 0x0123456789abcdefull ^ sink;
	const uint32_t d32 = 0x01234567u ^ sink;
	if (OP == OP_RD_B32 || OP == OP_RD_U8 || OP == OP_RD_U16) {
#define RD(ins)                                                                                                          \
		asm volatile(ins " %0, %8\n\t" ins " %1, %9\n\t" ins " %2, %10\n\t" ins " %3, %11\n\t" ins " %4, %12\n\t"         \
		             ins " %5, %13\n\t" ins " %6, %14\n\t" ins " %7, %15\n\ts_waitcnt lgkmcnt(0)"                         \
		             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)             \
		             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory")
		if (OP == OP_RD_B32) RD("ds_read_b32");
		else if (OP == OP_RD_U8) RD("ds_read_u8");
		else RD("ds_read_u16");
#undef RD
		sink ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
	} else if (OP == OP_RD_B64 || OP == OP_RD2_B32) {
#define RD(ins, suf)                                                                                                     \
		asm volatile(ins " %0, %8" suf "\n\t" ins " %1, %9" suf "\n\t" ins " %2, %10" suf "\n\t" ins " %3, %11" suf "\n\t" \
		             ins " %4, %12" suf "\n\t" ins " %5, %13" suf "\n\t" ins " %6, %14" suf "\n\t" ins " %7, %15" suf      \
		             "\n\ts_waitcnt lgkmcnt(0)"                                                                             \
		             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)             \
		             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory")
		if (OP == OP_RD_B64) RD("ds_read_b64", "");
		else RD("ds_read2_b32", " offset1:1");
#undef RD
		const uint64_t x = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
		sink ^= (uint32_t)x ^ (uint32_t)(x >> 32);
	} else if (OP == OP_RD_B128) {
		typedef uint32_t u4 __attribute__((ext_vector_type(4)));
		u4 v0, v1, v2, v3, v4, v5, v6, v7;
		asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
		             "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\t"
		             "s_waitcnt lgkmcnt(0)"
		             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
		             : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
		sink ^= v0.x ^ v1.y ^ v2.z ^ v3.w ^ v4.x ^ v5.y ^ v6.z ^ v7.w;
	} else if (OP == OP_WR_B32 || OP == OP_WR_B8 || OP == OP_WR_B16 || OP == OP_OR_B32) {
#define WR(ins)                                                                                                          \
		asm volatile(ins " %0, %8\n\t" ins " %1, %8\n\t" ins " %2, %8\n\t" ins " %3, %8\n\t" ins " %4, %8\n\t"            \
		             ins " %5, %8\n\t" ins " %6, %8\n\t" ins " %7, %8\n\ts_waitcnt lgkmcnt(0)"                            \
		             :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(d32) : "memory")
		if (OP == OP_WR_B32) WR("ds_write_b32");
		else if (OP == OP_WR_B8) WR("ds_write_b8");
		else if (OP == OP_WR_B16) WR("ds_write_b16");
		else WR("ds_or_b32");
#undef WR
	} else if (OP == OP_WR_B64) {
		asm volatile("ds_write_b64 %0, %8\n\tds_write_b64 %1, %8\n\tds_write_b64 %2, %8\n\tds_write_b64 %3, %8\n\t"
		             "ds_write_b64 %4, %8\n\tds_write_b64 %5, %8\n\tds_write_b64 %6, %8\n\tds_write_b64 %7, %8\n\t"
		             "s_waitcnt lgkmcnt(0)"
		             :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(d64) : "memory");
	} else if (OP == OP_WR_B128) {
		typedef uint32_t u4 __attribute__((ext_vector_type(4)));
		const u4 d = { d32, d32 + 1, d32 + 2, d32 + 3 };
		asm volatile("ds_write_b128 %0, %8\n\tds_write_b128 %1, %8\n\tds_write_b128 %2, %8\n\tds_write_b128 %3, %8\n\t"
		             "ds_write_b128 %4, %8\n\tds_write_b128 %5, %8\n\tds_write_b128 %6, %8\n\tds_write_b128 %7, %8\n\t"
		             "s_waitcnt lgkmcnt(0)"
		             :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(d) : "memory");
	} else if (OP == OP_WR2_B32) {
		asm volatile("ds_write2_b32 %0, %8, %9 offset1:1\n\tds_write2_b32 %1, %8, %9 offset1:1\n\t"
		             "ds_write2_b32 %2, %8, %9 offset1:1\n\tds_write2_b32 %3, %8, %9 offset1:1\n\t"
		             "ds_write2_b32 %4, %8, %9 offset1:1\n\tds_write2_b32 %5, %8, %9 offset1:1\n\t"
		             "ds_write2_b32 %6, %8, %9 offset1:1\n\tds_write2_b32 %7, %8, %9 offset1:1\n\t"
		             "s_waitcnt lgkmcnt(0)"
		             :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(d32), "v"(d32 + 1) : "memory");
	} else if (OP == OP_MSKOR_B32) {
		const uint32_t m = 0x00ffff00u;
		asm volatile("ds_mskor_b32 %0, %8, %9\n\tds_mskor_b32 %1, %8, %9\n\tds_mskor_b32 %2, %8, %9\n\tds_mskor_b32 %3, %8, %9\n\t"
		             "ds_mskor_b32 %4, %8, %9\n\tds_mskor_b32 %5, %8, %9\n\tds_mskor_b32 %6, %8, %9\n\tds_mskor_b32 %7, %8, %9\n\t"
		             "s_waitcnt lgkmcnt(0)"
		             :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(m), "v"(d32 & m) : "memory");
	} else if (OP == OP_MSKOR_B64) {
		const uint64_t m = 0x00ffffffffffff00ull;
		asm volatile("ds_mskor_b64 %0, %8, %9\n\tds_mskor_b64 %1, %8, %9\n\tds_mskor_b64 %2, %8, %9\n\tds_mskor_b64 %3, %8, %9\n\t"
		             "ds_mskor_b64 %4, %8, %9\n\tds_mskor_b64 %5, %8, %9\n\tds_mskor_b64 %6, %8, %9\n\tds_mskor_b64 %7, %8, %9\n\t"
		             "s_waitcnt lgkmcnt(0)"
		             :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(m), "v"(d64 & m) : "memory");
	}
}

/* align: address modulo `gran` is forced to `rem`; pattern 0 = random, 1 = lane-linear (stride = gran) */
template <int OP>
__global__ __launch_bounds__(1024) void k_lds(uint32_t gran, uint32_t rem, uint32_t active, uint32_t pattern,
    unsigned long long *cyc, uint32_t *sink_out)
{
	__shared__ __attribute__((aligned(16))) uint8_t win[65536 + 64];
	const uint32_t tid = threadIdx.x, lane = tid & 63;
	for (uint32_t i = tid; i < (65536 + 64) / 4; i += blockDim.x)
		((uint32_t *)win)[i] = i * 2654435761u;
	__syncthreads();
	uint32_t a[8];
	uint32_t x = (blockIdx.x * 1024u + tid) * 2654435761u + 12345u;
	const uint32_t base = (uint32_t)(uintptr_t)win;	/* LDS byte address */
#pragma unroll
	for (int j = 0; j < 8; j++) {
		x = x * 1664525u + 1013904223u;
		uint32_t off = pattern ? ((lane + 64u * (uint32_t)j + 512u * (tid >> 6)) * gran) & 0xFFFFu : (x >> 8) & 0xFFFFu;
		off = off / gran * gran + rem;
		if (off > 65536u - 16u) off -= 4096u;
		a[j] = base + off;
	}
	uint32_t sink = tid;
	const bool on = lane < active;
	__syncthreads();
	const unsigned long long t0 = __builtin_readcyclecounter();
	if (on) {
		for (int it = 0; it < ITERS; it++)
			group8<OP>(a, sink);
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0)
		cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
	if (sink == 0x12345u)
		sink_out[0] = sink;
}

typedef void (*kern_t)(uint32_t, uint32_t, uint32_t, uint32_t, unsigned long long *, uint32_t *);
template <int OP> static kern_t get() { return k_lds<OP>; }

int main()
{
	kern_t ks[OP_COUNT] = { get<OP_RD_B32>(), get<OP_RD_B64>(), get<OP_RD2_B32>(), get<OP_RD_B128>(), get<OP_RD_U8>(), get<OP_RD_U16>(),
		get<OP_WR_B32>(), get<OP_WR_B64>(), get<OP_WR2_B32>(), get<OP_WR_B8>(), get<OP_WR_B16>(), get<OP_MSKOR_B32>(), get<OP_MSKOR_B64>(), get<OP_OR_B32>(), get<OP_WR_B128>() };
	unsigned long long *d_cyc; uint32_t *d_sink;
	const int nblk = 256;
	hipMalloc(&d_cyc, nblk * 16 * sizeof(unsigned long long));
	hipMalloc(&d_sink, 4);
	std::vector<unsigned long long> h(nblk * 16);
	struct cfg { int op; uint32_t gran, rem; } cfgs[] = {
		{ OP_RD_B32, 4, 0 }, { OP_RD_B32, 4, 1 }, { OP_RD_B32, 4, 2 },
		{ OP_RD_B64, 8, 0 }, { OP_RD_B64, 8, 4 }, { OP_RD_B64, 8, 1 }, { OP_RD_B64, 8, 2 },
		{ OP_RD2_B32, 4, 0 }, { OP_RD2_B32, 8, 0 },
		{ OP_RD_B128, 16, 0 }, { OP_RD_B128, 16, 8 }, { OP_RD_B128, 16, 4 },
		{ OP_RD_U8, 1, 0 }, { OP_RD_U16, 2, 0 }, { OP_RD_U16, 2, 1 },
		{ OP_WR_B32, 4, 0 }, { OP_WR_B32, 4, 1 }, { OP_WR_B32, 4, 2 },
		{ OP_WR_B64, 8, 0 }, { OP_WR_B64, 8, 4 }, { OP_WR_B64, 8, 1 },
		{ OP_WR2_B32, 4, 0 }, { OP_WR2_B32, 8, 0 },
		{ OP_WR_B8, 1, 0 }, { OP_WR_B16, 2, 0 }, { OP_WR_B16, 2, 1 },
		{ OP_MSKOR_B32, 4, 0 }, { OP_MSKOR_B64, 8, 0 }, { OP_OR_B32, 4, 0 },
		{ OP_WR_B128, 16, 0 }, { OP_WR_B128, 16, 8 },
	};
	printf("%-14s %4s %3s %4s %5s %6s | cycles per wave-instruction on one CU (pipe time), single-wave latency\n", "op", "gran", "rem", "pat", "lanes", "waves");
	for (auto &c : cfgs) {
		for (uint32_t pattern = 0; pattern < 2; pattern++) {
			for (uint32_t active : { 64u, 16u, 4u }) {
				if (pattern == 1 && active != 64u) continue;
				double res[2];
				int wi = 0;
				for (uint32_t waves : { 1u, 16u }) {
					hipMemset(d_cyc, 0, nblk * 16 * sizeof(unsigned long long));
					for (int rep = 0; rep < 2; rep++)
						hipLaunchKernelGGL(ks[c.op], dim3(nblk), dim3(64 * waves), 0, 0, c.gran, c.rem, active, pattern, d_cyc, d_sink);
					hipDeviceSynchronize();
					hipMemcpy(h.data(), d_cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
					/* median over blocks of the slowest wave of the block */
					std::vector<double> per;
					for (int b = 0; b < nblk; b++) {
						unsigned long long mx = 0;
						for (uint32_t w = 0; w < waves; w++) mx = h[b * 16 + w] > mx ? h[b * 16 + w] : mx;
						per.push_back((double)mx);
					}
					std::sort(per.begin(), per.end());
					res[wi++] = per[nblk / 2] / (double)(ITERS * 8) / (double)waves;
				}
				printf("%-14s %4u %3u %4s %5u | 1 wave: %7.1f cyc/instr   16 waves: %6.2f cyc/instr (CU pipe)\n",
				    op_name[c.op], c.gran, c.rem, pattern ? "lin" : "rand", active, res[0], res[1]);
			}
		}
	}
	hipError_t e = hipGetLastError();
	printf("last error: %s\n", hipGetErrorString(e));
	return e == hipSuccess ? 0 : 1;
}
