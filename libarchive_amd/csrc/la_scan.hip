/*
 * la_scan.hip -- exclusive prefix sum of per-block decoded lengths (packs the
 * blocks back to back in the decoded slab) and the batch summary reduction
 * (first failing event in stream order, which the host maps to the reference's
 * return code and error string -- SURVEY section 5 "failure detection").
 */
#include "la_dev.h"

#define SCAN_TPB   256
#define SCAN_ITEMS 8
#define SCAN_TILE  (SCAN_TPB * SCAN_ITEMS)

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint64_t t = __shfl_up(v, d, 64);
		if (lane >= d) v += t;
	}
	return v;
}

/* block-wide exclusive scan of one u64 per thread; returns the exclusive value, *total = block sum */
__device__ uint64_t block_excl_scan(uint64_t v, uint64_t *total)
{
	__shared__ uint64_t wsum[SCAN_TPB / 64];
	int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint64_t inc = wave_incl_scan_u64(v, lane);
	if (lane == 63) wsum[w] = inc;
	__syncthreads();
	uint64_t base = 0, tot = 0;
	for (int k = 0; k < SCAN_TPB / 64; k++) {
		if (k < w) base += wsum[k];
		tot += wsum[k];
	}
	__syncthreads();
	*total = tot;
	return base + inc - v;
}

__global__ __launch_bounds__(SCAN_TPB) void scan_tile_sums(const uint32_t *__restrict__ in, uint32_t n,
    uint64_t *__restrict__ tile_sums)
{
	uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
	uint64_t s = 0;
	for (int k = 0; k < SCAN_ITEMS; k++)
		if (base + k < n) s += in[base + k];
	uint64_t tot;
	block_excl_scan(s, &tot);
	if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_TPB) void scan_tile_offsets(uint64_t *tile_sums, uint32_t ntiles)
{
	uint64_t carry = 0;
	for (uint32_t b0 = 0; b0 < ntiles; b0 += SCAN_TPB) {
		uint32_t i = b0 + threadIdx.x;
		uint64_t v = i < ntiles ? tile_sums[i] : 0, tot;
		uint64_t ex = block_excl_scan(v, &tot);
		if (i < ntiles) tile_sums[i] = carry + ex;
		carry += tot;
	}
}

__global__ __launch_bounds__(SCAN_TPB) void scan_write(const uint32_t *__restrict__ in, uint32_t n,
    const uint64_t *__restrict__ tile_offs, uint64_t *out, const uint64_t *base_ptr)
{
	/* base_ptr may alias out[0] (a slice continuing the previous slice's scan): read it first */
	const uint64_t base0 = base_ptr ? *base_ptr : 0;
	uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
	uint32_t v[SCAN_ITEMS];
	uint64_t s = 0;
	for (int k = 0; k < SCAN_ITEMS; k++) {
		v[k] = base + k < n ? in[base + k] : 0u;
		s += v[k];
	}
	uint64_t tot;
	uint64_t ex = block_excl_scan(s, &tot) + tile_offs[blockIdx.x] + base0;
	for (int k = 0; k < SCAN_ITEMS; k++) {
		if (base + k < n) out[base + k] = ex;
		ex += v[k];
		if (base + k + 1 == n) out[n] = ex;
	}
}

uint64_t la_scan_scratch_bytes(uint32_t n)
{
	return ((uint64_t)(n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(uint64_t);
}

void la_launch_scan_u32(hipStream_t s, const uint32_t *d_in, uint32_t n, uint64_t *d_out, void *d_scratch)
{
	la_launch_scan_u32_base(s, d_in, n, d_out, d_scratch, NULL);
}

/* d_base: device pointer to the value the scan starts from (NULL = 0); may be d_out itself */
void la_launch_scan_u32_base(hipStream_t s, const uint32_t *d_in, uint32_t n, uint64_t *d_out, void *d_scratch,
    const uint64_t *d_base)
{
	if (n == 0) {
		if (!d_base)
			(void)hipMemsetAsync(d_out, 0, sizeof(uint64_t), s);
		else if (d_base != d_out)
			(void)hipMemcpyAsync(d_out, d_base, sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
		return;
	}
	uint32_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
	uint64_t *tiles = (uint64_t *)d_scratch;
	hipLaunchKernelGGL(scan_tile_sums, dim3(ntiles), dim3(SCAN_TPB), 0, s, d_in, n, tiles);
	hipLaunchKernelGGL(scan_tile_offsets, dim3(1), dim3(SCAN_TPB), 0, s, tiles, ntiles);
	hipLaunchKernelGGL(scan_write, dim3(ntiles), dim3(SCAN_TPB), 0, s, d_in, n, tiles, d_out, d_base);
}

/* ------------------------------------------------------------------ summary */

__global__ void summary_init(la_batch_summary *sm)
{
	sm->total_out = 0;
	sm->n_bad_units = 0;
	sm->n_bad_frames = 0;
	sm->first_bad_unit = 0xFFFFFFFFu;
	sm->first_bad_frame = 0xFFFFFFFFu;
	sm->first_zero_unit = 0xFFFFFFFFu;
	sm->reserved = 0;
}

__global__ __launch_bounds__(256) void summary_reduce(const uint32_t *__restrict__ out_len,
    const uint32_t *__restrict__ ustatus, uint32_t n_units, const uint32_t *__restrict__ fstatus,
    uint32_t n_frames, const uint64_t *__restrict__ dst_off, la_batch_summary *sm)
{
	uint32_t stride = gridDim.x * blockDim.x;
	uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t bad = 0, first_bad = 0xFFFFFFFFu, first_zero = 0xFFFFFFFFu;
	for (uint32_t i = t; i < n_units; i += stride) {
		if (ustatus[i] != LA_ST_OK) {
			bad++;
			if (i < first_bad) first_bad = i;
		} else if (out_len[i] == 0 && i < first_zero)
			first_zero = i;
	}
	uint32_t fbad = 0, first_fbad = 0xFFFFFFFFu;
	for (uint32_t i = t; i < n_frames; i += stride) {
		if (fstatus[i] != LA_ST_OK) {
			fbad++;
			if (i < first_fbad) first_fbad = i;
		}
	}
	/* wave-level reduction, then one atomic per wave */
	for (int d = 32; d >= 1; d >>= 1) {
		bad += __shfl_down(bad, d, 64);
		fbad += __shfl_down(fbad, d, 64);
		first_bad = min(first_bad, (uint32_t)__shfl_down(first_bad, d, 64));
		first_zero = min(first_zero, (uint32_t)__shfl_down(first_zero, d, 64));
		first_fbad = min(first_fbad, (uint32_t)__shfl_down(first_fbad, d, 64));
	}
	if ((threadIdx.x & 63) == 0) {
		if (bad) atomicAdd(&sm->n_bad_units, bad);
		if (fbad) atomicAdd(&sm->n_bad_frames, fbad);
		if (first_bad != 0xFFFFFFFFu) atomicMin(&sm->first_bad_unit, first_bad);
		if (first_zero != 0xFFFFFFFFu) atomicMin(&sm->first_zero_unit, first_zero);
		if (first_fbad != 0xFFFFFFFFu) atomicMin(&sm->first_bad_frame, first_fbad);
	}
	if (t == 0)
		sm->total_out = dst_off ? dst_off[n_units] : 0;
}

void la_launch_lz4_summary(hipStream_t s, const uint32_t *d_out_len, const uint32_t *d_block_status,
    uint32_t n_blocks, const uint32_t *d_frame_status, uint32_t n_frames,
    const uint64_t *d_dst_off, la_batch_summary *d_summary)
{
	hipLaunchKernelGGL(summary_init, dim3(1), dim3(1), 0, s, d_summary);
	uint32_t n = n_blocks > n_frames ? n_blocks : n_frames;
	uint32_t blocks = (n + 255) / 256;
	if (blocks == 0) blocks = 1;
	if (blocks > 1024) blocks = 1024;
	hipLaunchKernelGGL(summary_reduce, dim3(blocks), dim3(256), 0, s, d_out_len, d_block_status,
	    n_blocks, d_frame_status, n_frames, d_dst_off, d_summary);
}
