"""The Arrow IPC parser of flight.hip under AddressSanitizer + UBSan on the CPU (host-only code, g++): the seeds are
pyarrow-written requests and ingest streams, the registry HAS the dataset (a stub index that touches every byte it is
handed), so mutated k values, list offsets, buffer lengths and row counts reach the code behind the NotFound check.
>= 10,000 mutations per run, zero sanitizer reports; plus the hand-patched LargeList offsets that used to walk out of
the message body (int64 overflow in the byte count) and a schema that exceeds the total-node budget."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32
DIM = 8


def _ipc(batch_or_batches, schema=None):
    batches = batch_or_batches if isinstance(batch_or_batches, list) else [batch_or_batches]
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, schema or batches[0].schema) as w:
        for b in batches:
            w.write_batch(b)
    return sink.getvalue().to_pybytes()


def _request(qtype, k=5, ktype=pa.int32(), dstype=pa.string(), extra=False):
    q = np.arange(DIM, dtype=F) + 1
    cols, names = [], []
    if extra:
        cols.append(pa.array([["x", "y"]], pa.list_(pa.string())))
        names.append("tags")
    cols += [pa.array(["ds"], dstype), pa.array([k], ktype), pa.array([64], pa.int32())]
    names += ["dataset", "k", "ef"]
    if qtype == "fsl":
        cols.append(pa.FixedSizeListArray.from_arrays(pa.array(q, pa.float32()), DIM))
    elif qtype == "list":
        cols.append(pa.array([q.tolist()], pa.list_(pa.float32())))
    else:
        cols.append(pa.array([q.tolist()], pa.large_list(pa.float32())))
    names.append("query_vector")
    return _ipc(pa.record_batch(cols, names=names))


def _ingest(rows=37, with_id=True, idtype=pa.uint64()):
    X = np.random.default_rng(3).random((rows, DIM), dtype=F)
    cols = [pa.FixedSizeListArray.from_arrays(pa.array(X.reshape(-1), pa.float32()), DIM)]
    names = ["vector"]
    if with_id:
        cols.append(pa.array(np.arange(rows) * 3 + 1, idtype))
        names.append("id")
    b = pa.record_batch(cols, names=names)
    return _ipc([b, b])


def _patched_large_list():
    """the advisor's reproduction: LargeList offsets a = 2^61, b = 2^61 + dim -- b * 4 overflows int64, the old length
    check passed and the query pointer left the body"""
    data = bytearray(_request("large_list"))
    good = struct.pack("<qq", 0, DIM)
    at = bytes(data).rfind(good)
    assert at > 0
    data[at:at + 16] = struct.pack("<qq", 1 << 61, (1 << 61) + DIM)
    return bytes(data)


def _huge_rows_ingest():
    """RecordBatch.length = 2^61 with an id column: used to throw std::length_error out of the C entry point"""
    data = bytearray(_ingest(rows=5))
    needle = struct.pack("<q", 5)
    hits = [i for i in range(0, len(data) - 8, 8) if data[i:i + 8] == needle]
    assert hits
    for at in hits:  # the batch length and the FieldNode lengths
        data[at:at + 8] = struct.pack("<q", 1 << 61)
    return bytes(data)


def _wide_schema():
    """more Field nodes than the reader's total budget (4096): rejected before it is materialised"""
    inner = pa.struct([pa.field(f"c{i}", pa.int32()) for i in range(70)])
    outer = pa.struct([pa.field(f"s{i}", inner) for i in range(70)])
    schema = pa.schema([pa.field("blob", outer), pa.field("dataset", pa.string())])
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, schema):
        pass
    return sink.getvalue().to_pybytes()


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    out = tmp_path_factory.mktemp("fuzz") / "flight_fuzz"
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-x", "c++", os.path.join(ROOT, "longbow_amd", "csrc", "flight.hip"), os.path.join(ROOT, "tests", "cpp", "flight_fuzz.cpp"),
           "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("libasan / libubsan not installed")
    assert r.returncode == 0, r.stderr
    return str(out)


def test_ipc_parser_survives_10k_mutations_under_asan_and_ubsan(fuzz_binary, tmp_path):
    seeds = []

    def put(kind, name, data):
        p = tmp_path / name
        p.write_bytes(data)
        seeds.append(f"{kind}:{p}")

    put("request", "fsl.ipc", _request("fsl"))
    put("request", "list.ipc", _request("list", ktype=pa.int64(), extra=True))
    put("request", "large.ipc", _request("large_list", dstype=pa.large_string()))
    put("ingest", "ing_u64.ipc", _ingest())
    put("ingest", "ing_noid.ipc", _ingest(with_id=False))
    put("ingest", "ing_u32.ipc", _ingest(idtype=pa.uint32()))
    put("must_fail_request", "patched_offsets.ipc", _patched_large_list())
    put("must_fail_request", "k_huge.ipc", _request("fsl", k=2**31 - 1))
    put("must_fail_request", "wide_schema.ipc", _wide_schema())
    put("must_fail_ingest", "rows_2p61.ipc", _huge_rows_ingest())
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_binary, "12000", "20261004", str(DIM)] + seeds, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    head = r.stdout.splitlines()[0]
    assert head.startswith("mutations: 12000")
    reached = int(head.split("searches reached:")[1].split()[0])
    assert reached > 200, head  # the mutations do get past NotFound and into the search call
