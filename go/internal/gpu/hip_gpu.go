//go:build gpu && linux

// hip_gpu.go -- cgo binding of liblongbow_gpu.so (hand-written HIP kernels for AMD MI355X /
// gfx950) behind Longbow's gpu.Index plug-point.
//
// Drop this file into internal/gpu/ of 23skdu/longbow in place of faiss_gpu.go and point
// gpu_enabled.go's NewIndexWithConfig at NewHIPIndex (see INTEGRATION.md).  It implements exactly
// the contract of internal/gpu/interface.go:3-19 and keeps the FAISS binding's behaviour
// (internal/gpu/faiss_gpu.go:44-167): validation before the C call, borrowed slices, int return
// codes turned into errors, RWMutex (Add/Close exclusive, Search shared), finalizer, metrics.
//
// NOTE: this build image has no Go toolchain, so this file is provided as reviewed source; the
// same C ABI is exercised end to end by the ctypes mirror in longbow_amd/gpu.py and tests/.
package gpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -llongbow_gpu
#include <stdlib.h>
#include "longbow_gpu.h"
*/
import "C"

import (
	"context"
	"errors"
	"fmt"
	"runtime"
	"sync"
	"time"
	"unsafe"

	"github.com/23skdu/longbow/internal/metrics"
	"github.com/23skdu/longbow/internal/simd"
)

// HIPIndex wraps an lb_gpu_index handle.
type HIPIndex struct {
	dim      int
	deviceID int
	metric   simd.MetricType
	h        *C.lb_gpu_index
	mu       sync.RWMutex
	closed   bool
}

// HIPConfig extends GPUConfig with the distance metric (simd.MetricType values 0/1/2 are the
// lb_metric values).  The zero value is Euclidean, what every live reference path uses.
type HIPConfig struct {
	GPUConfig
	Metric simd.MetricType
}

func hipError(h *C.lb_gpu_index, op string, rc C.int) error {
	msg := C.GoString(C.lb_gpu_status_string(rc))
	if h != nil {
		if detail := C.GoString(C.lb_gpu_last_error(h)); detail != "" {
			msg += ": " + detail
		}
	}
	if rc == C.LB_ERR_NO_DEVICE {
		return fmt.Errorf("%s: %w", op, ErrGPUNotAvailable)
	}
	return fmt.Errorf("GPU index %s failed with code %d (%s)", op, int(rc), msg)
}

// ErrGPUNotAvailable mirrors the !gpu stub (stub.go:10) so callers can errors.Is on it.
var ErrGPUNotAvailable = errors.New("GPU support not enabled in this build")

// NewHIPIndex creates a GPU index on an MI355X.  Replaces NewFaissGPUIndex (faiss_gpu.go:44-72).
func NewHIPIndex(cfg GPUConfig) (Index, error) {
	return NewHIPIndexWithMetric(HIPConfig{GPUConfig: cfg})
}

func NewHIPIndexWithMetric(cfg HIPConfig) (Index, error) {
	if cfg.Dimension <= 0 {
		return nil, fmt.Errorf("dimension must be positive, got %d", cfg.Dimension)
	}
	var st C.int
	h := C.lb_gpu_index_new(C.int(cfg.DeviceID), C.int(cfg.Dimension), C.int(cfg.Metric), &st)
	if h == nil {
		return nil, hipError(nil, "init", st)
	}
	idx := &HIPIndex{dim: cfg.Dimension, deviceID: cfg.DeviceID, metric: cfg.Metric, h: h}
	runtime.SetFinalizer(idx, (*HIPIndex).Close)
	return idx, nil
}

// Add appends vectors (row-major []float32, e.g. the values buffer of an Arrow
// FixedSizeList<float32> column).  The library copies during the call (pinned staging + DMA)
// and retains no Go pointer.
func (idx *HIPIndex) Add(ids []int64, vectors []float32) error {
	idx.mu.Lock()
	defer idx.mu.Unlock()
	if idx.closed {
		return fmt.Errorf("index is closed")
	}
	if len(vectors)%idx.dim != 0 {
		return fmt.Errorf("vector data length %d not divisible by dimension %d", len(vectors), idx.dim)
	}
	n := len(vectors) / idx.dim
	if len(ids) != n {
		return fmt.Errorf("id count %d does not match vector count %d", len(ids), n)
	}
	if n == 0 {
		return nil
	}
	rc := C.lb_gpu_index_add(idx.h, C.int64_t(n),
		(*C.float)(unsafe.Pointer(&vectors[0])), (*C.int64_t)(unsafe.Pointer(&ids[0])))
	if rc != C.LB_OK {
		metrics.VectorSearchGPUOperationsTotal.WithLabelValues("add", "error").Inc()
		return hipError(idx.h, "add", rc)
	}
	metrics.VectorSearchGPUOperationsTotal.WithLabelValues("add", "success").Inc()
	return nil
}

// Search returns the k nearest neighbours of one query, ascending by distance.
func (idx *HIPIndex) Search(vector []float32, k int) ([]int64, []float32, error) {
	ids, dist, err := idx.SearchBatch(vector, 1, k)
	return ids, dist, err
}

// SearchBatch searches nq queries (row-major) at once; results are nq*k, query-major.
// Fewer than k hits are padded by the library with label -1 / MaxFloat32; the single-query
// Search trims the padding so callers see min(k, N) results like BruteForceIndex does
// (internal/store/adaptive_index.go:215-222).
func (idx *HIPIndex) SearchBatch(queries []float32, nq, k int) ([]int64, []float32, error) {
	idx.mu.RLock()
	defer idx.mu.RUnlock()
	if idx.closed {
		return nil, nil, fmt.Errorf("index is closed")
	}
	if len(queries) != nq*idx.dim {
		return nil, nil, fmt.Errorf("query vector dimension %d does not match index dimension %d", len(queries)/max(nq, 1), idx.dim)
	}
	if k <= 0 || nq <= 0 {
		return nil, nil, nil
	}
	distances := make([]float32, nq*k)
	labels := make([]int64, nq*k)
	start := time.Now()
	rc := C.lb_gpu_index_search(idx.h, C.int64_t(nq), (*C.float)(unsafe.Pointer(&queries[0])), C.int(k),
		(*C.float)(unsafe.Pointer(&distances[0])), (*C.int64_t)(unsafe.Pointer(&labels[0])))
	if rc != C.LB_OK {
		metrics.VectorSearchGPUOperationsTotal.WithLabelValues("search", "error").Inc()
		return nil, nil, hipError(idx.h, "search", rc)
	}
	metrics.VectorSearchGPULatencySeconds.WithLabelValues("search").Observe(time.Since(start).Seconds())
	metrics.VectorSearchGPUOperationsTotal.WithLabelValues("search", "success").Inc()
	if nq == 1 {
		n := k
		for n > 0 && labels[n-1] < 0 {
			n--
		}
		return labels[:n], distances[:n], nil
	}
	return labels, distances, nil
}

// SearchBatchContext is SearchBatch under a context.Context: the reference's brute-force loop polls ctx.Err() every
// 1000 rows (internal/store/adaptive_index.go:182); here the context's cancellation and deadline are handed to the
// library through an lb_cancel, which it polls before every kernel launch of the search (one launch covers at most
// one pass over <= 2.5M rows).  Returns ctx.Err() when the search was cut short.
func (idx *HIPIndex) SearchBatchContext(ctx context.Context, queries []float32, nq, k int) ([]int64, []float32, error) {
	if err := ctx.Err(); err != nil {
		return nil, nil, err
	}
	idx.mu.RLock()
	defer idx.mu.RUnlock()
	if idx.closed {
		return nil, nil, fmt.Errorf("index is closed")
	}
	if nq <= 0 || k <= 0 {
		return nil, nil, nil
	}
	if len(queries) != nq*idx.dim {
		return nil, nil, fmt.Errorf("query vector dimension %d does not match index dimension %d", len(queries)/max(nq, 1), idx.dim)
	}
	cc := C.lb_cancel_new()
	if cc == nil {
		return nil, nil, fmt.Errorf("out of memory")
	}
	if dl, ok := ctx.Deadline(); ok {
		C.lb_cancel_set_deadline_ms(cc, C.int64_t(max(time.Until(dl).Milliseconds(), 0)))
	}
	done := make(chan struct{})
	exited := make(chan struct{})
	go func() { // fires the library-side flag when the context ends before the call does
		defer close(exited)
		select {
		case <-ctx.Done():
			C.lb_cancel_fire(cc)
		case <-done:
		}
	}()
	// The token is freed only after the watcher has LEFT: a caller's `defer cancel()` fires right after this function
	// returns, the watcher may then pick ctx.Done() over done (select is random when both are ready), and a fire on a
	// freed token would be a store into freed memory.  One deferred function, in this order: stop, join, free.
	defer func() {
		close(done)
		<-exited
		C.lb_cancel_free(cc)
	}()
	distances := make([]float32, nq*k)
	labels := make([]int64, nq*k)
	rc := C.lb_gpu_index_search_ctx(idx.h, C.int64_t(nq), (*C.float)(unsafe.Pointer(&queries[0])), C.int(k),
		(*C.float)(unsafe.Pointer(&distances[0])), (*C.int64_t)(unsafe.Pointer(&labels[0])), cc)
	switch rc {
	case C.LB_OK:
		return labels, distances, nil
	case C.LB_ERR_CANCELLED:
		return nil, nil, context.Canceled
	case C.LB_ERR_DEADLINE:
		return nil, nil, context.DeadlineExceeded
	}
	metrics.VectorSearchGPUOperationsTotal.WithLabelValues("search", "error").Inc()
	return nil, nil, hipError(idx.h, "search", rc)
}

// SetCandidateMode chooses how batched searches generate candidates (results are identical in every mode):
// C.LB_CAND_AUTO (default), C.LB_CAND_F32_MFMA (strict), C.LB_CAND_SPLIT_BF16 (corpus image, second copy in HBM),
// C.LB_CAND_SPLIT_BF16_INREG.
func (idx *HIPIndex) SetCandidateMode(mode int) error {
	idx.mu.Lock()
	defer idx.mu.Unlock()
	if idx.closed {
		return fmt.Errorf("index is closed")
	}
	if rc := C.lb_gpu_index_set_candidate_mode(idx.h, C.int(mode)); rc != C.LB_OK {
		return hipError(idx.h, "set_candidate_mode", rc)
	}
	return nil
}

// SetF16Image controls the fp16 copy of the corpus the single-product route reads (half the corpus's bytes again, kept only
// while the device has room: lb_gpu_index_set_f16_image).  true is the default; false trades ~20 % of large-batch throughput
// for the memory.  Results do not depend on it.
func (idx *HIPIndex) SetF16Image(on bool) error {
	idx.mu.Lock()
	defer idx.mu.Unlock()
	if idx.closed {
		return fmt.Errorf("index is closed")
	}
	m := C.int(0)
	if on {
		m = 1
	}
	if rc := C.lb_gpu_index_set_f16_image(idx.h, m); rc != C.LB_OK {
		return hipError(idx.h, "set_f16_image", rc)
	}
	return nil
}

// SetSearchCombining controls whether concurrent Search calls (one query each, one goroutine each) are answered by ONE batched
// device search when they overlap (lb_gpu_index_set_search_combining; default on).  The lists are identical either way.
func (idx *HIPIndex) SetSearchCombining(on bool) error {
	idx.mu.Lock()
	defer idx.mu.Unlock()
	if idx.closed {
		return fmt.Errorf("index is closed")
	}
	m := C.int(0)
	if on {
		m = 1
	}
	if rc := C.lb_gpu_index_set_search_combining(idx.h, m); rc != C.LB_OK {
		return hipError(idx.h, "set_search_combining", rc)
	}
	return nil
}

// F16ImageBytes reports the HBM that copy holds right now (0: none) -- for the memory gauge.
func (idx *HIPIndex) F16ImageBytes() int64 {
	idx.mu.RLock()
	defer idx.mu.RUnlock()
	if idx.closed {
		return 0
	}
	return int64(C.lb_gpu_index_f16_image_bytes(idx.h))
}

// FusedGiveups reports how many small-batch searches had their in-launch threshold hand-off give up (~1 ms) and were
// redone on the exact path since the index was created: a latency event for a metrics gauge.
func (idx *HIPIndex) FusedGiveups() int64 {
	idx.mu.RLock()
	defer idx.mu.RUnlock()
	if idx.closed {
		return 0
	}
	return int64(C.lb_gpu_index_fused_giveups(idx.h))
}

// SetFilter installs a byte-per-row predicate mask (0 = excluded), e.g. the output of
// query.FilterEvaluator (internal/query/filter_evaluator.go:79-115).  nil clears it.
func (idx *HIPIndex) SetFilter(mask []byte) error {
	idx.mu.Lock()
	defer idx.mu.Unlock()
	if idx.closed {
		return fmt.Errorf("index is closed")
	}
	var p *C.uint8_t
	if len(mask) > 0 {
		p = (*C.uint8_t)(unsafe.Pointer(&mask[0]))
	}
	if rc := C.lb_gpu_index_set_filter(idx.h, p, C.int64_t(len(mask))); rc != C.LB_OK {
		return hipError(idx.h, "set_filter", rc)
	}
	return nil
}

// Close releases HBM; idempotent.
func (idx *HIPIndex) Close() error {
	idx.mu.Lock()
	defer idx.mu.Unlock()
	if idx.closed {
		return nil
	}
	if idx.h != nil {
		C.lb_gpu_index_free(idx.h)
		idx.h = nil
	}
	idx.closed = true
	return nil
}

// Rerank is the distance step of processChunkInternal (internal/store/parallel_search.go:274-364) for
// candidates that are already resident on the GPU: rows are row positions (the labels Search returns when
// Add got no ids), dist[i] is what simd.EuclideanDistanceBatchFlat would give for the gathered vectors
// (4-accumulator order), score[i] = 1/(1+dist[i]).  No vectors cross PCIe.
func (idx *HIPIndex) Rerank(query []float32, rows []int64) (dist, score []float32, err error) {
	idx.mu.RLock()
	defer idx.mu.RUnlock()
	if idx.closed {
		return nil, nil, fmt.Errorf("index is closed")
	}
	if len(query) != idx.dim {
		return nil, nil, fmt.Errorf("query vector dimension %d does not match index dimension %d", len(query), idx.dim)
	}
	if len(rows) == 0 {
		return nil, nil, nil
	}
	dist = make([]float32, len(rows))
	score = make([]float32, len(rows))
	rc := C.lb_gpu_index_rerank(idx.h, (*C.float)(unsafe.Pointer(&query[0])), (*C.int64_t)(unsafe.Pointer(&rows[0])),
		C.int64_t(len(rows)), C.int(C.LB_ORDER_UNROLL4), (*C.float)(unsafe.Pointer(&dist[0])), (*C.float)(unsafe.Pointer(&score[0])))
	if rc != C.LB_OK {
		return nil, nil, hipError(idx.h, "rerank", rc)
	}
	return dist, score, nil
}

// NodeIndex shards a corpus over every GPU of the node from ONE process (the Longbow server): one HIPIndex
// per device, one RCCL communicator (lb_gpu_comm_init_all, SURVEY 8b's lb_gpu_comm_init(ndev)).  Search is
// ShardedHNSW.SearchVectors' fan-out + concat + sort (internal/store/sharded_hnsw.go:414-503) done as shard
// searches + one all-gather over xGMI + a device merge.  Which ids a shard holds is the caller's choice
// (store.RingSharder, internal/store/sharding_strategy.go:40-127): Add takes the shard explicitly.
type NodeIndex struct {
	shards []*HIPIndex
	comm   *C.lb_gpu_comm
	mu     sync.RWMutex
}

func NewNodeIndex(devices []int, cfg HIPConfig) (*NodeIndex, error) {
	if len(devices) == 0 {
		return nil, fmt.Errorf("no devices given")
	}
	n := &NodeIndex{}
	cdev := make([]C.int, len(devices))
	for i, d := range devices {
		c := cfg
		c.DeviceID = d
		ix, err := NewHIPIndexWithMetric(c)
		if err != nil {
			n.Close()
			return nil, err
		}
		n.shards = append(n.shards, ix.(*HIPIndex))
		cdev[i] = C.int(d)
	}
	var st C.int
	n.comm = C.lb_gpu_comm_init_all(C.int(len(devices)), &cdev[0], &st)
	if n.comm == nil {
		n.Close()
		return nil, hipError(nil, "comm init", st)
	}
	return n, nil
}

// Prepare sizes the exchange buffers for searches of up to nqMax queries and kMax results (lb_gpu_comm_prepare): after it
// a search within that size allocates nothing between the shard searches and the exchange.
func (n *NodeIndex) Prepare(nqMax, kMax int) error {
	n.mu.Lock()
	defer n.mu.Unlock()
	if n.comm == nil {
		return fmt.Errorf("index is closed")
	}
	if rc := C.lb_gpu_comm_prepare(n.comm, C.int64_t(nqMax), C.int(kMax)); rc != C.LB_OK {
		return fmt.Errorf("comm prepare failed with code %d (%s)", int(rc), C.GoString(C.lb_gpu_comm_last_error(n.comm)))
	}
	return nil
}

// Add appends vectors to one shard (ids are the global VectorIDs; they travel with the results).
func (n *NodeIndex) Add(shard int, ids []int64, vectors []float32) error {
	if shard < 0 || shard >= len(n.shards) {
		return fmt.Errorf("shard %d out of range", shard)
	}
	return n.shards[shard].Add(ids, vectors)
}

// SearchBatch returns the global top-k of nq queries (row-major), nq*k results, query-major.
func (n *NodeIndex) SearchBatch(queries []float32, nq, k int) ([]int64, []float32, error) {
	n.mu.RLock()
	defer n.mu.RUnlock()
	if n.comm == nil {
		return nil, nil, fmt.Errorf("index is closed")
	}
	if nq <= 0 || k <= 0 {
		return nil, nil, nil
	}
	if dim := n.shards[0].dim; len(queries) != nq*dim { // (checked before &queries[0] goes to C)
		return nil, nil, fmt.Errorf("query vector dimension %d does not match index dimension %d", len(queries)/nq, dim)
	}
	hs := make([]*C.lb_gpu_index, len(n.shards))
	for i, s := range n.shards {
		hs[i] = s.h
	}
	distances := make([]float32, nq*k)
	labels := make([]int64, nq*k)
	rc := C.lb_gpu_comm_search_all(n.comm, &hs[0], C.int64_t(nq), (*C.float)(unsafe.Pointer(&queries[0])), C.int(k),
		(*C.float)(unsafe.Pointer(&distances[0])), (*C.int64_t)(unsafe.Pointer(&labels[0])))
	if rc != C.LB_OK {
		return nil, nil, fmt.Errorf("sharded search failed with code %d (%s)", int(rc), C.GoString(C.lb_gpu_comm_last_error(n.comm)))
	}
	return labels, distances, nil
}

func (n *NodeIndex) Close() error {
	n.mu.Lock()
	defer n.mu.Unlock()
	if n.comm != nil {
		C.lb_gpu_comm_free(n.comm)
		n.comm = nil
	}
	for _, s := range n.shards {
		s.Close()
	}
	n.shards = nil
	return nil
}
