"""Process-wide side streams.

HIP multiplexes streams over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); a stream is bound to the
least-used queue when it is first used and keeps its reference for as long as it exists -- and torch never destroys the
streams of its pool.  A second FacePipeline (or a second bench leg) built on fresh `torch.cuda.Stream()` objects can
therefore land its detection and embedding streams on ONE queue that dead streams left under-counted, and the two
stages silently serialise (measured: 79 k -> 64 k faces/s for the second pipeline of a process, 234 k -> 200 k
embeddings/s for an embed leg after a pipeline leg).  The package hands out the same few stream objects for the life of
the process instead: role i of a device is always the same stream, on the queue it got when nothing else was attached.
"""
import torch

_streams = {}


def side_stream(device, index):
    """The process-wide side stream number `index` (0, 1, 2, ...) of `device`."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise ValueError("side_stream: a cuda device is required")
    di = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (di, int(index))
    if key not in _streams:
        # default priority: measured at bs 256 x 3 lanes, raising any lane's priority (or all of them) costs 15-25 %
        _streams[key] = torch.cuda.Stream(device=torch.device("cuda", di))
    return _streams[key]


def side_streams(device, count, first=0):
    return [side_stream(device, first + i) for i in range(count)]
