//go:build gpu && linux

// hip_gpu_test.go -- restates internal/gpu/gpu_test.go:24-46 against the HIP binding and adds the
// context-lifetime case of SearchBatchContext.  Run on an MI355X box with
//
//	go test -tags gpu -race ./internal/gpu/
//
// NOTE: this build image has no Go toolchain; the file is reviewed source like hip_gpu.go.  The same
// scenario (a token fired from a second thread right after the call returned, then freed) runs against
// the C ABI in tests/test_gpu_hardening.py::test_cancel_token_fire_races_the_return.
package gpu

import (
	"context"
	"testing"
	"time"
)

func newTestIndex(t *testing.T, n, dim int) *HIPIndex {
	t.Helper()
	idx, err := NewHIPIndex(GPUConfig{DeviceID: 0, Dimension: dim})
	if err != nil {
		t.Skipf("GPU not available: %v", err)
	}
	vectors := make([]float32, n*dim)
	ids := make([]int64, n)
	for i := 0; i < n; i++ {
		ids[i] = int64(i)
		for j := 0; j < dim; j++ {
			vectors[i*dim+j] = float32(i) * 0.01
		}
	}
	if err := idx.Add(ids, vectors); err != nil {
		t.Fatalf("Add: %v", err)
	}
	return idx.(*HIPIndex)
}

// gpu_test.go:24-46: the nearest neighbour of a stored vector is itself.
func TestHIPIndexSelfQuery(t *testing.T) {
	idx := newTestIndex(t, 10, 128)
	defer idx.Close()
	q := make([]float32, 128)
	ids, dist, err := idx.Search(q, 5)
	if err != nil {
		t.Fatalf("Search: %v", err)
	}
	if len(ids) != 5 || ids[0] != 0 || dist[0] > 0.01 {
		t.Fatalf("got ids %v dist %v", ids, dist)
	}
}

// The idiomatic caller: ctx, cancel := context.WithTimeout(...); defer cancel().  cancel() fires
// immediately after SearchBatchContext returns, while its watcher goroutine may still be choosing
// between ctx.Done() and done.  The token must outlive the watcher (-race / ASan would report the
// store into freed memory that the round-3 ordering allowed).
func TestSearchBatchContextCancelRightAfterReturn(t *testing.T) {
	idx := newTestIndex(t, 4096, 128)
	defer idx.Close()
	q := make([]float32, 128)
	for i := 0; i < 2000; i++ {
		ctx, cancel := context.WithTimeout(context.Background(), time.Minute)
		ids, _, err := idx.SearchBatchContext(ctx, q, 1, 5)
		cancel() // races the watcher's exit
		if err != nil || len(ids) != 5 {
			t.Fatalf("iteration %d: ids %v err %v", i, ids, err)
		}
	}
}

func TestSearchBatchContextCanceledBeforeCall(t *testing.T) {
	idx := newTestIndex(t, 64, 128)
	defer idx.Close()
	ctx, cancel := context.WithCancel(context.Background())
	cancel()
	if _, _, err := idx.SearchBatchContext(ctx, make([]float32, 128), 1, 5); err != context.Canceled {
		t.Fatalf("want context.Canceled, got %v", err)
	}
}
