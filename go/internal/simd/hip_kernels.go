//go:build gpu && linux

// hip_kernels.go -- plugs the HIP batch kernels of liblongbow_gpu.so into internal/simd's own
// dispatch surface: the KernelRegistry (registry.go:82-124), DispatchDistance's lookup rule
// (dispatch.go:264-302: exact dims first, then the generic dims = 0 entry) and the metric names of
// internal/core/enums.go:6-13.
//
// Drop this file into internal/simd/ of 23skdu/longbow (build tag gpu).  Nothing existing changes:
//   - the per-pair kernels registered by dispatch.go:221-234 stay where they are (one pair per PCIe
//     round trip would be absurd on a GPU);
//   - the HIP kernels are BATCH kernels (one query x n rows of a flat buffer), registered under their own
//     key dims = BatchFlatDims so that Registry.Get's "exact match first" rule finds them and a build
//     without this file falls back to the generic per-pair kernel in a loop (DispatchBatchFlat below);
//   - EuclideanDistanceBatchFlat's call sites (internal/store/parallel_search.go:347) can switch to
//     DispatchBatchFlat(MetricEuclidean, ...) and get the GPU when it is there.
//
// NOTE: this build image has no Go toolchain, so this file is provided as reviewed source; the same C
// entry points and the same registry rule are exercised by longbow_amd/simd.py and tests/.
package simd

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -llongbow_gpu
#include "longbow_gpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math"
	"unsafe"
)

// BatchFlatDims is the KernelKey.Dims under which batch-flat kernels are registered.  No vector has a
// negative length, so DispatchDistance (dims = len(a)) can never be handed one of these.
const BatchFlatDims = -1

// BatchFlatFunc is the signature of simd.EuclideanDistanceBatchFlat (batch_operations.go:64-87).
type BatchFlatFunc func(query, flatVectors []float32, numVectors, dims int, results []float32) error

// HIPDevice is the GPU the batch kernels run on (LONGBOW_GPU_DEVICE_ID, cmd/longbow/main.go:99).
var HIPDevice = 0

// HIPAccumulationOrder selects which of the reference's two f32 summation orders the kernels
// reproduce bit for bit: 1 = four accumulators (euclideanUnrolled4x, what BatchFlat runs,
// simd.go:365-396), 0 = one accumulator (the generic / test-oracle order, simd.go:138-163).
var HIPAccumulationOrder = 1

func hipBatchFlat(metric MetricType) BatchFlatFunc {
	return func(query, flat []float32, n, dims int, results []float32) error {
		if n == 0 {
			return nil // batch_operations.go:65-67
		}
		if len(results) != n {
			return errors.New("simd: results length mismatch") // simd.go:204-206
		}
		if len(flat) < n*dims {
			return errors.New("simd: flatVectors too small") // simd.go:207-209
		}
		if len(query) != dims {
			return errors.New("simd: query dimension mismatch") // simd.go:215-217
		}
		rc := C.lb_simd_distance_batch_flat(C.int(HIPDevice), C.int(metric), C.int(HIPAccumulationOrder),
			(*C.float)(unsafe.Pointer(&query[0])), (*C.float)(unsafe.Pointer(&flat[0])),
			C.int64_t(n), C.int(dims), (*C.float)(unsafe.Pointer(&results[0])))
		if rc != C.LB_OK {
			return fmt.Errorf("simd: HIP batch kernel failed: %s (code %d)", C.GoString(C.lb_gpu_status_string(rc)), int(rc))
		}
		return nil
	}
}

func init() {
	if C.lb_gpu_device_count() <= 0 {
		return // no GPU: the CPU kernels registered by dispatch.go stay the only ones
	}
	Registry.Register(MetricEuclidean, DataTypeFloat32, BatchFlatDims, hipBatchFlat(MetricEuclidean))
	Registry.Register(MetricCosine, DataTypeFloat32, BatchFlatDims, hipBatchFlat(MetricCosine))
	Registry.Register(MetricDotProduct, DataTypeFloat32, BatchFlatDims, hipBatchFlat(MetricDotProduct))
}

// DispatchBatchFlat is DispatchDistance's rule for one query against n rows of a flat buffer:
// Registry.Get(metric, float32, BatchFlatDims) finds the HIP batch kernel when this file registered one;
// otherwise Get falls back to the generic per-pair kernel (dims = 0), which is applied row by row --
// exactly what the reference's generic batch loops do (simd.go:185-267).  DotProduct results are the RAW
// dot products, as simd.DotProductBatch returns them (batch_operations.go:146-157).
func DispatchBatchFlat(metric MetricType, query, flat []float32, n, dims int, results []float32) error {
	kernel := Registry.Get(metric, DataTypeFloat32, BatchFlatDims)
	switch k := kernel.(type) {
	case BatchFlatFunc:
		return k(query, flat, n, dims, results)
	case distanceFunc:
		if len(results) < n || len(flat) < n*dims {
			return errors.New("simd: results slice too small")
		}
		for i := 0; i < n; i++ {
			d, err := k(query, flat[i*dims:(i+1)*dims])
			if err != nil {
				d = math.MaxFloat32 // batch_operations.go:39-42
			}
			results[i] = d
		}
		return nil
	case nil:
		return fmt.Errorf("simd: no kernel found for %s/%s", metric, DataTypeFloat32)
	default:
		return fmt.Errorf("simd: invalid kernel type for %s: %T", DataTypeFloat32, kernel)
	}
}

// MetricFromCore maps internal/core's metric strings (core/enums.go:6-13: "euclidean", "cosine",
// "dot_product") -- and MetricType.String()'s own "dot" (registry.go:17-28) -- onto MetricType, whose
// values are the lb_metric values of the C ABI.  core.DistanceMetric is a string type, so callers pass
// string(metric); this keeps internal/simd free of an import cycle with internal/core.
func MetricFromCore(name string) (MetricType, error) {
	switch name {
	case "euclidean", "":
		return MetricEuclidean, nil // the zero value: every live reference path is Euclidean
	case "cosine":
		return MetricCosine, nil
	case "dot_product", "dot":
		return MetricDotProduct, nil
	default:
		return 0, fmt.Errorf("simd: unknown distance metric %q", name)
	}
}
