"""longbow_amd -- MI355X (gfx950) k-NN distance / top-k / PQ-ADC backend for Longbow.

Host-side mirror (Python, over ctypes) of the two reference interfaces the HIP
library drops in behind:

  longbow_amd.gpu   <->  internal/gpu   (Index{Add, Search, Close}, GPUConfig, NewIndex*)
  longbow_amd.simd  <->  internal/simd  (MetricType, *DistanceBatch*, ADCDistanceBatch)
  longbow_amd.pq    <->  internal/pq    (BuildADCTable, ADCDistanceBatch, codebook blob)
  longbow_amd.sharded    RingSharder partition + RCCL all-gather merge (one process per GPU)
  longbow_amd.arrow_io   Arrow RecordBatch ingestion + DoExchange / DoAction("VectorSearch") framing
  longbow_amd.hybrid     ReciprocalRankFusion + the GPU -> HNSW candidate hand-off rule

Everything computes in liblongbow_gpu.so (hand-written HIP).  There is no CPU
fallback: importing is cheap, but any compute call raises if the library or a GPU
is missing.
"""
from . import _lib  # noqa: F401
from .simd import MetricType  # noqa: F401

__all__ = ["gpu", "simd", "pq", "sharded", "arrow_io", "hybrid", "MetricType"]
