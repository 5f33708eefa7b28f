// knn_host.h -- the host side of the HOST entry points (trx_index_add, trx_index_search): the caller's NumPy-style array
// becomes device rows.  The reference hands FAISS int64 count vectors and int8 bit vectors (retrieve/retrieve_faiss.py:24-27,
// 36-44, 66, 71); FAISS' wrapper turns them into float32 on one thread.  Here a pool of worker threads narrows integer rows
// to int8 when every value fits (else to the float32 FAISS would have seen) straight into pinned staging buffers, one chunk
// crossing PCIe while the next is narrowed.  No arithmetic of the search happens here: only the change of storage type.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace trx {

// what the staged rows are on the device
enum { STAGED_F32 = 0, STAGED_BF16 = 1, STAGED_I8 = 2 };

// bytes per component of a host dtype (include/trx_knn.h: TRX_DTYPE_*), 0 if unknown
int host_dtype_size(int dtype);
// is it one of the integer types the narrowing pass takes (I64, I32, I16, U8)?
bool host_dtype_is_wide_int(int dtype);

// Stage rows [0, m) of `src` (row-major, d components each, host dtype `dtype`) into device memory `dev` on `copy_stream`.
//   *staged (in / out): STAGED_I8 on entry asks for the narrow form; integer rows that do not fit a signed byte come back as
//   STAGED_F32 (the whole block is staged again: `dev` must hold m * d * 4 bytes in that case -- see staged_capacity).
// Returns hipSuccess with every copy enqueued on copy_stream (the caller synchronises it or waits on an event), or the error.
// Serialised inside: one staging pipeline per process.
hipError_t stage_host_rows(const void* src, int dtype, int64_t m, int d, void* dev, int* staged, hipStream_t copy_stream);

// device bytes `dev` must offer for a block of m rows of host dtype `dtype` when asked for form `staged`
size_t staged_bytes(int64_t m, int d, int staged);

// copy m * bytes_per_row from pinned-staged device results back into pageable host memory with the same pipeline (device ->
// pinned -> worker threads -> dst); waits for everything before returning
hipError_t unstage_to_host(void* dst, const void* dev, size_t bytes, hipStream_t copy_stream);

// the conversion alone, host memory to host memory, on the pool (what the pipeline runs per chunk): count elements of `dtype`
// -> `staged` form; 1 when the narrow form was asked for and some value does not fit a signed byte (dst is then garbage)
int convert_host_rows(const void* src, int dtype, int64_t count, void* dst, int staged);

// worker threads the pool runs (TRX_HOST_THREADS, default min(cores of the affinity mask, 32))
int host_threads();

}  // namespace trx
