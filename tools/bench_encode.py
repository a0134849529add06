#!/usr/bin/env python
"""Device-side PostingsSerializer throughput: tq_encode_postings_device over the
postings of the bench's synthetic 10M-doc segment, inputs and output resident in HBM.  One JSON
line: postings/s, the roofline of the measure+scan+write kernels (algorithmic bytes = 9 B per
posting read — doc id, tf, fieldnorm id — + the bytes written) and the oracle's serializer timed on one host core.
The output is checked byte for byte against the segment the oracle serialised."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    from oracle import oracle as O  # workload generator, checker, cpu baseline
    import tantivy_amd
    from tantivy_amd.binding import lib, _check

    seg = O.synth_segment(args.docs, n_terms=256, with_positions=False)
    starts, docs, tfs = [0], [], []
    for t in range(len(seg.terms)):
        d, f = O.decode_postings(seg, t)
        docs.append(d)
        tfs.append(f)
        starts.append(starts[-1] + len(d))
    ts = np.array(starts, np.uint64)
    docs, tfs = np.concatenate(docs), np.concatenate(tfs)
    n_post = int(ts[-1])
    avg = float(np.float32(seg.total_num_tokens) / np.float32(seg.max_doc))
    idx_len = getattr(seg, "idx_len", len(seg.idx))
    want = np.asarray(seg.idx[8:idx_len])

    enc = tantivy_amd.Encoder(0)
    d_ts = torch.from_numpy(ts.view(np.int64)).cuda()
    d_docs = torch.from_numpy(docs.view(np.int32)).cuda()
    d_tfs = torch.from_numpy(tfs.view(np.int32)).cuda()
    d_fn = torch.from_numpy(np.ascontiguousarray(seg.fieldnorm)).cuda()
    cap = int(want.size) + 1024
    d_out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    d_ots = torch.zeros(len(ts), dtype=torch.int64, device="cuda")
    need = C.c_uint64()
    stream = torch.cuda.Stream()

    def step():
        _check(lib().tq_encode_postings_device(
            enc.raw, len(ts) - 1, ts.ctypes.data, d_ts.data_ptr(), d_docs.data_ptr(),
            d_tfs.data_ptr(), d_fn.data_ptr(), seg.max_doc, C.c_float(avg), seg.record_option,
            d_out.data_ptr(), cap, d_ots.data_ptr(), C.byref(need), stream.cuda_stream))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    kms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        kms.append(enc.last_kernel_ms())  # HIP events on the stream the kernels run on
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    got = d_out[: need.value].cpu().numpy()
    ots = d_ots.cpu().numpy().view(np.uint64)
    assert need.value == want.size and np.array_equal(got, want), "device bytes != serializer bytes"
    for t, ti in enumerate(seg.terms):
        assert (int(ots[t]), int(ots[t + 1])) == (ti.postings_start, ti.postings_end)

    cpu = None
    if not args.no_cpu_baseline:
        sub = min(len(ts) - 1, 256)
        t1 = time.perf_counter()
        body, _ = O.serialize_postings_batch(ts[: sub + 1], docs, tfs, seg.fieldnorm, seg.max_doc, avg,
                                             seg.record_option)
        wall = time.perf_counter() - t1
        assert np.array_equal(body, want[: body.size])
        cpu = {"value": round(int(ts[sub]) / wall, 1), "unit": "postings/s", "cores": 1, "kind": "port",
               "sample": "the same %d posting lists through the oracle's PostingsSerializer "
                         "restatement, one thread, %.2f s" % (sub, wall)}
    k = float(np.mean(kms))
    algo = 9 * n_post + int(want.size)  # doc + tf + fieldnorm id read per posting, bytes written
    print(json.dumps({
        "metric": "postings_encoded_per_sec", "value": round(n_post * args.steps / elapsed, 1),
        "unit": "postings/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "dtype": "u32+f32", "data": "synthetic",
        "config": {"workload": "encode: %d postings of %d terms (10M-doc Zipf segment, WithFreqs, "
                               "block-max metadata), inputs and output in HBM" % (n_post, len(ts) - 1),
                   "output_bytes": int(want.size)},
        "roofline": {"bound": "hbm", "achieved": round(algo / (k * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(algo / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel": "enc_measure + scan + enc_write",
                     "kernel_ms_avg": round(k, 4), "algorithmic_bytes_per_launch": algo},
        "cpu_baseline": cpu, "bytes_equal_serializer": True}))
    enc.close()


if __name__ == "__main__":
    main()
