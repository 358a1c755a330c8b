"""Two batches in flight on two HIP streams.

The proposal stages (flatten, top-k, decode, NMS, gather) are a few latency-bound kernels with one
workgroup per image; RoIAlign fills the chip and is bound by its output stream.  Run one after the
other on a single stream the chip idles through the first and the proposal kernels of the NEXT batch
could have used that time.  `ProposalRoIPipeline.submit` enqueues the proposal stages on one stream
and RoIAlign on a second one behind an event, so that consecutive submits overlap.  The reference runs
everything on one CPU thread (RPN/rpn.py:46-85 then ROIHead/roi_head.py:127-130); this is host-side
scheduling only -- kernels and results are the ones of the serial path.

Beside a RoIAlign launch every CU's LDS is taken, so a proposal kernel that needs a CU of its own waits
for one: keep those few -- `ops.topk_desc(..., max_slices=1)` / `rpn.topk_max_slices = 1` (one workgroup
per image instead of the split top-k's one per CU: 18.2k against 16.0k images/s in bench.py).
"""
import torch


class ProposalRoIPipeline:
    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("ProposalRoIPipeline needs a HIP device (there is no CPU path)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.s_prop = torch.cuda.Stream(device=self.device)
        self.s_roi = torch.cuda.Stream(device=self.device)

    def submit(self, proposal_fn, roi_fn):
        """proposal_fn() -> (props, n_keep) runs on the proposal stream; roi_fn(props, n_keep) -> anything
        runs on the RoI stream once the proposals are ready.  Returns (props, n_keep, roi_result); nothing
        is synchronised -- call `synchronize()` (or wait on the streams) before reading results on
        another stream.  Inputs the callables read must already be complete on these streams
        (call `wait_for_current()` after producing them on the current stream)."""
        with torch.cuda.stream(self.s_prop):
            props, n_keep = proposal_fn()
            ready = torch.cuda.Event()
            ready.record(self.s_prop)
        with torch.cuda.stream(self.s_roi):
            self.s_roi.wait_event(ready)
            # the caching allocator must not hand these blocks out again while RoIAlign still reads them
            props.record_stream(self.s_roi)
            n_keep.record_stream(self.s_roi)
            result = roi_fn(props, n_keep)
        return props, n_keep, result

    def wait_for_current(self):
        """Both streams wait for everything enqueued so far on the caller's current stream."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.s_prop.wait_event(ev)
        self.s_roi.wait_event(ev)

    def join_current(self):
        """The caller's current stream waits for both pipeline streams (no host synchronisation)."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.s_prop)
        cur.wait_stream(self.s_roi)

    def synchronize(self):
        self.s_prop.synchronize()
        self.s_roi.synchronize()
