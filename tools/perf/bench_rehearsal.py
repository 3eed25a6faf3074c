"""MEMB_BENCH_REHEARSAL=cpu: the multi-rank PLUMBING of bench.py on a machine without a GPU (8 ranks in the CPU test
suite: launcher, rendezvous, barriers, the max-over-ranks reduce, all_gather_object of the per-rank summaries, the
strong-scaling split, the JSON line). Never set by the driver."""
import time


def install_host_stand_ins(torch, memb_amd):
    """MEMB_BENCH_REHEARSAL=cpu: the multi-rank PLUMBING of this script on a machine without a GPU (8 ranks in
    the CPU test suite: launcher, rendezvous, barriers, the max-over-ranks reduce, all_gather_object of the
    per-rank summaries, the strong-scaling split, the JSON line). Everything that would touch the device is
    replaced by a host stand-in -- wall-clock `events`, tensors in host memory, a Reader on the product's host
    path (device='cpu', the reference's own serial / threaded decode restated) -- so the numbers such a run
    prints are NOT measurements of anything; the line says so in `rehearsal`. Never set by the driver."""
    import numpy as np

    class Event:
        def __init__(self, enable_timing=True):
            self.at = 0.0

        def record(self):
            self.at = time.perf_counter()

        def elapsed_time(self, other):
            return max((other.at - self.at) * 1e3, 1e-6)

    class Stream:
        cuda_stream = 0

    torch.cuda.Event = Event
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: Stream()
    torch.Tensor.cuda = lambda self, *a, **k: self

    def on_host(function):
        def wrapped(*args, **kwargs):
            if str(kwargs.get('device', '')).startswith('cuda'):
                kwargs['device'] = 'cpu'
            kwargs.pop('pin_memory', None)
            return function(*args, **kwargs)
        return wrapped

    for name in ('empty', 'zeros', 'full', 'tensor', 'arange'):
        setattr(torch, name, on_host(getattr(torch, name)))

    product_reader = memb_amd.Reader

    class HostReader(product_reader):
        def __init__(self, filename, num_threads=0, device=None, **kwargs):
            super().__init__(filename, num_threads, device='cpu', **kwargs)

        def rows_embedding_device(self, rows, out=None, col_off=0, accumulate=False, divisor=0.0, order=None):
            ids = np.ascontiguousarray(rows.numpy()).view(np.uint32)
            if out is None:
                out = torch.empty((len(ids), self.dim), dtype=torch.float32)
            self.rows_embedding_into(ids, out.numpy(), col_off)
            return out

    memb_amd.Reader = HostReader
