// rocBLAS handle shared by the translation units that call plain library GEMMs (train_enc.hip, w2v_enc.hip).
#pragma once
#include <rocblas/rocblas.h>
#include "ts_common.hpp"

namespace ts {

// One handle per process (one process per GPU); bound to the caller's stream on every use.
inline int blas(hipStream_t stream, rocblas_handle* h) {
  static rocblas_handle handle = nullptr;
  if (!handle && rocblas_create_handle(&handle) != rocblas_status_success) return TS_EUNSUPPORTED;
  if (rocblas_set_stream(handle, stream) != rocblas_status_success) return TS_EUNSUPPORTED;
  *h = handle;
  return TS_OK;
}

}  // namespace ts
