"""Throughput mode of the codec: consecutive batches software-pipelined over two HIP streams.

A codec step is g_a (MFMA-bound) -> prior competition and rANS encoder (latency-bound: a few thousand serial
chains) -> byte strings on the host -> rANS decoder (latency-bound) -> g_s (MFMA-bound).  Run one batch after the
other, the GPU idles through the latency-bound parts and the host round trip in the middle: a fifth of the step.  The
batches are independent, so this driver keeps two in flight, each on its own stream, and orders the work so that

    stream of batch n     : g_a(n) | competition, encoder(n) | host: bytes of n | ........ wait ....... | decoder(n) | g_s(n)
    stream of batch n - 1 : decoder(n-1) | wait | g_s(n-1) .....................| g_a(n+1) ............................

the two transforms never run beside each other (g_s(n-1) waits for an event behind g_a(n); g_a(n+1) is enqueued behind
g_s(n-1) on the same stream: two MFMA-bound kernels that share the CUs only slow each other, measured 12 % worse than
no pipelining at all; and this order is also a matter of correctness -- with the first layers of g_s(n-1) beside the last
layer of g_a(n), a few steps per thousand had a wrong tile in that layer: an LDS-DMA kernel must not share a CU with
another transform kernel, see lds DMA notes in csrc/conv_common.h), the encoder runs beside g_s(n-1), and the decoder of a batch, held back until the g_s before it
has ended, beside the first layer of the next g_a (beside the second layer instead it costs the dominant kernel 9 % of
its rate for 0.6 % of throughput).  Every batch still goes
encode -> bytes on the host -> decode; only the order in which the host enqueues the phases changes
(encode_batch_begin/finish, decode_batch_begin/synthesis/finish of the model).  The reference codes one image at a
time on one stream (src/compression/tests.py:439-460); it has no counterpart of this driver.
"""
import torch

from . import ops


class CodecPipeline:
    def __init__(self, model, device, after_synthesis=None):
        """after_synthesis(job, strings, tag) -> anything: called right behind g_s with the batch's stream current
        (job["recon"]: the reconstruction; with a target given to submit() it is clamped and job["sse"] holds the
        per-image squared errors); e.g. the batch's statistics.  Its value comes back with the result."""
        self.model = model
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)]
        self.after_synthesis = after_synthesis
        self.pending = None          # (byte strings, idx, h, w, stream, tag, target) of the batch that is encoded and not yet decoded
        self.count = 0
        self.synthesis_done = None   # event behind the g_s (and hook) enqueued last
        for s in self.streams:       # whatever produced the inputs on the caller's stream comes first
            s.wait_stream(torch.cuda.current_stream(self.device))

    def _decode(self, undecoded, transforms_done):
        """Enqueues decoder and g_s of the batch whose byte strings are on the host, each behind its event."""
        strings, idx, h, w, stream, tag, target = undecoded
        with torch.cuda.stream(stream):
            if self.synthesis_done is not None:
                stream.wait_event(self.synthesis_done)           # decoder(n-1) behind g_s(n-2): beside the first layer of g_a(n)
            job = self.model.decode_batch_begin(strings, idx, h, w)
            if transforms_done is not None:
                stream.wait_event(transforms_done)               # g_s(n-1) behind g_a(n)
            self.model.decode_batch_synthesis(job, score_against=target)
            extra = self.after_synthesis(job, strings, tag) if self.after_synthesis is not None else None
            self.synthesis_done = torch.cuda.Event()
            self.synthesis_done.record()
        return job, stream, strings, tag, extra

    def submit(self, x, tag=None, target=None):
        """Encodes batch n (its byte strings reach the host inside this call) and decodes batch n - 1.  target: the
        image batch n's reconstruction is to be scored against (see decode_batch_synthesis).  -> None for the first
        batch, else (byte strings, x_hat, after_synthesis' value, tag) of batch n - 1; x_hat is complete on the stream
        `result_stream` (the caller's next use must be ordered behind it)."""
        stream = self.streams[self.count & 1]
        idle = self.streams[(self.count + 1) & 1]                # (its batch's encoder has ended: encode_batch_finish waited)
        self.count += 1
        # The image goes to the first layer's layout on the idle stream, beside the g_s that `stream` is still running:
        # enqueued on `stream` the 45 us pass would sit between that g_s and this batch's g_a, on the critical path.
        idle.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(idle):
            x4 = ops.nchw_to_nhwc(x.contiguous(), 4)
            laid_out = torch.cuda.Event()
            laid_out.record()
        x4.record_stream(stream)
        with torch.cuda.stream(stream):
            stream.wait_event(laid_out)
            enc = self.model.encode_batch_begin(x, input_nhwc=x4)
        done = None
        if self.pending is not None:
            done = self._decode(self.pending, enc["transforms_done"])
        with torch.cuda.stream(stream):
            strings, idx, h, w = self.model.encode_batch_finish(enc)
        self.pending = (strings, idx, h, w, stream, tag, target)
        return self._finish(done)

    def _finish(self, done):
        if done is None:
            return None
        job, stream, strings, tag, extra = done
        with torch.cuda.stream(stream):
            recon = self.model.decode_batch_finish(job)
        self.result_stream = stream
        return strings, recon, extra, tag

    def flush(self):
        """Decodes the batch still in flight (-> its result, or None if there is none)."""
        if self.pending is None:
            return None
        done = self._decode(self.pending, None)
        self.pending = None
        return self._finish(done)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()
