"""Where the gradient buckets of the data-parallel exchange become ready inside the backward: one GPU, the real
executor backward at the BASELINE size, `BucketedAllReduce` driven by the executor's range callback exactly as in a multi-rank
run, with the collective replaced by a same-size device-to-device copy on the SAME side stream (one GPU has no peer).  HIP events:
  ready_ms  = position of the compute stream when bucket k was flushed (the event the side stream waits on)
  xfer_*    = start / end of the stand-in transfer on the side stream
  bwd_ms    = end of the whole backward on the compute stream
Overlap by construction = every bucket but the last is ready (and its transfer done) long before bwd_ms.  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opticalflowdiffusion_amd import FlowDiffuser, parallel as P   # noqa: E402


class TracedSync(P.BucketedAllReduce):
    def __init__(self, bucket_bytes):
        super().__init__(bucket_bytes)
        self.trace = []
        self.scratch = None

    def _world(self):
        return 2                                       # take the multi-rank code path

    def _flush(self, flat):
        ranges, self._pending, self._pending_floats = self._pending, [], 0
        if not ranges:
            return
        self._done.extend(ranges)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(flat.device))
        self._comm.wait_event(ev)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self._comm):
            s0.record(self._comm)
            for b, e in ranges:                        # stand-in for dist.all_reduce(flat[b:e]): the same bytes move once
                self.scratch[b:e].copy_(flat[b:e])
            s1.record(self._comm)
        self.trace.append((sum(e - b for b, e in ranges) * 4, len(ranges), ev, s0, s1))

    def finish(self, flat):
        self._flush(flat)
        torch.cuda.current_stream(flat.device).wait_stream(self._comm)
        return self._done


def main():
    B, H, W = 16, 440, 1024
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    fd = FlowDiffuser(dict(target="flow", image_size=[H, W], timesteps=1000, flow_max=20, zero_init=False)).to(dev)
    fd.log_dict = lambda *a, **k: None
    sync = TracedSync(32 << 20)
    fd.unet.grad_sync = sync
    img = torch.rand(B, 3, H, W, device=dev)
    flow = (torch.rand(B, 2, H, W, device=dev) * 2 - 1) * 10
    for it in range(3):
        for p in fd.model.parameters():
            p.grad = None
        loss = fd.training_step((img, img, flow), it)
        if sync.scratch is None:
            sync.scratch = torch.empty_like(fd.unet.flat_grads())
        sync.trace = []
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        loss.backward()
        t1.record()
        torch.cuda.synchronize()
    bwd = t0.elapsed_time(t1)
    buckets = [{"bytes": nb, "ranges": nr, "ready_ms": round(t0.elapsed_time(ev), 3), "xfer_start_ms": round(t0.elapsed_time(s0), 3),
                "xfer_end_ms": round(t0.elapsed_time(s1), 3)} for nb, nr, ev, s0, s1 in sync.trace]
    print(json.dumps({"what": "gradient buckets inside the backward (one GPU, stand-in transfer on the side stream)", "batch": B, "size": [H, W],
                      "backward_ms": round(bwd, 3), "bucket_bytes": 32 << 20, "buckets": buckets,
                      "ready_before_backward_end": sum(1 for b in buckets if b["xfer_end_ms"] < bwd), "n_buckets": len(buckets)}))


if __name__ == "__main__":
    main()
