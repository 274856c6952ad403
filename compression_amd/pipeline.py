"""Throughput mode of the codec: consecutive batches software-pipelined over two HIP streams.

A codec step is g_a (MFMA-bound) -> prior competition and rANS encoder (latency-bound: a few thousand serial
chains) -> byte strings on the host -> rANS decoder (latency-bound) -> g_s (MFMA-bound).  Run one batch after the
other, the GPU idles through the latency-bound parts and the host round trip in the middle: a fifth of the step.  The
batches are independent, so this driver keeps two in flight, each on its own stream, and orders the work so that

    stream of batch n     : g_a(n) | competition, encoder(n) | ............ wait ............ | decoder(n) | g_s(n)
    stream of batch n - 1 : decoder(n-1) | wait | g_s(n-1) ..................| g_a(n+1) ............................
    host                  : enqueue g_a(n), encoder(n) | fetch the bytes of n-1 | enqueue decoder(n-1), g_s(n-1) | n+1 ...

the bulk of the two transforms never runs side by side (g_s(n-1) waits for an event behind the third layer of g_a(n),
manynets_compressor.SYNTHESIS_MAY_START_BEHIND_LAYER, so only the last analysis layer -- too small to fill the chip --
shares the CUs with the first transposed layers: +1.5 %; g_a(n+1) is enqueued behind g_s(n-1) on the same stream.  Two
MFMA-bound kernels that share the CUs for long only slow each other: measured 12 % worse than no pipelining at all),
the encoder runs beside g_s(n-1), and the decoder of a batch, held back until the g_s before it has ended, beside the
first layer of the next g_a (both kernels are built to fit a SIMD together: 2 x 208 + 2 x 48 registers).  The host
fetches a batch's byte strings one call after it enqueued that batch's encoder, so it never waits for the GPU with
nothing queued behind.  Every batch still goes encode -> bytes on the host -> decode; only the order in which the
host enqueues the phases changes (encode_batch_begin/finish, decode_batch_begin/synthesis/finish of the model).

early_first_layer=True additionally starts the first layer of g_a(n+1) beside the third layer of g_s(n-1) (256 + 208
registers fit a SIMD, and alone each kernel keeps the matrix pipe under half busy).  Measured, the pair then takes
2.55 ms instead of 1.43 + 1.24 ms one after the other, and the decoder, which loses its place beside the first
layer, costs the second layer 0.28 ms: 1.3-2.5 % slower over the whole step, hence off by default.

The driver works with any model that splits its codec into encode_batch_begin / encode_batch_finish (-> a tuple: the
byte strings first, the whole tuple being the arguments of) decode_batch_begin / decode_batch_synthesis /
decode_batch_finish: the Manypriors models (one string per image, coded under the winning prior's table) and the
hyperprior models (two strings per image; their encode_batch_finish also enqueues the two coders, which then run beside
the next batch's g_a).

The reference codes one image at a time on one stream (src/compression/tests.py:439-460); it has no counterpart of
this driver.
"""
import torch

from . import ops


class CodecPipeline:
    def __init__(self, model, device, after_synthesis=None, early_first_layer=False, decoder_behind="synthesis"):
        """after_synthesis(job, strings, tag) -> anything: called right behind g_s with the batch's stream current
        (job["recon"]: the reconstruction; with a target given to submit() it is clamped and job["sse"] holds the
        per-image squared errors); e.g. the batch's statistics.  Its value comes back with the result.
        early_first_layer: the first layer of g_a(n+1) runs beside the third layer of g_s(n-1) (see the module's text).
        decoder_behind: what the decoder of a batch waits for, "synthesis" (the end of the g_s before it) or "deconv2"
        (the second layer of that g_s)."""
        if decoder_behind not in ("synthesis", "deconv2"):
            raise ValueError("decoder_behind: 'synthesis' or 'deconv2'")
        self.model = model
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)]
        self.early = torch.cuda.Stream(self.device)   # image layout and first layer of every batch
        self.early_first_layer = early_first_layer
        self.decoder_behind = decoder_behind
        self.after_synthesis = after_synthesis
        self.encoding = None         # (encoder job, stream, tag, target) of the batch whose byte strings are not fetched yet
        self.count = 0
        self.synthesis_done = None   # event behind the g_s (and hook) enqueued last
        self.deconv2_done = None     # event in front of the third layer of that g_s
        for s in self.streams:       # whatever produced the inputs on the caller's stream comes first
            s.wait_stream(torch.cuda.current_stream(self.device))

    def _decode(self, undecoded, transforms_done):
        """Enqueues decoder and g_s of the batch whose byte strings are on the host, each behind its event."""
        coded, stream, tag, target = undecoded                  # coded: what encode_batch_finish returned, strings first
        strings = coded[0]
        with torch.cuda.stream(stream):
            held_by = self.synthesis_done if self.decoder_behind == "synthesis" else self.deconv2_done
            if held_by is not None:
                stream.wait_event(held_by)
            job = self.model.decode_batch_begin(*coded)
            if transforms_done is not None:
                stream.wait_event(transforms_done)               # g_s(n-1) behind the third layer of g_a(n)
            events = {}
            self.model.decode_batch_synthesis(job, score_against=target, events=events)
            extra = self.after_synthesis(job, strings, tag) if self.after_synthesis is not None else None
            self.synthesis_done = torch.cuda.Event()
            self.synthesis_done.record()
            self.deconv2_done = events.get("deconv2_done")
        return job, stream, strings, tag, extra

    def _fetch_and_decode(self, transforms_done):
        enc, stream, tag, target = self.encoding
        self.encoding = None
        with torch.cuda.stream(stream):
            coded = self.model.encode_batch_finish(enc)
        return self._decode((coded, stream, tag, target), transforms_done)

    def submit(self, x, tag=None, target=None):
        """Enqueues the encoder of batch n, then fetches the byte strings of batch n - 1 and enqueues its decoder and
        g_s.  target: the image batch n's reconstruction is to be scored against (see decode_batch_synthesis).  ->
        None for the first batch, else (byte strings, x_hat, after_synthesis' value, tag) of batch n - 1; x_hat is
        complete on the stream `result_stream` (the caller's next use must be ordered behind it)."""
        stream = self.streams[self.count & 1]
        self.count += 1
        # The image goes to the first layer's layout on a stream of its own, beside whatever runs: enqueued on `stream`
        # the 45 us pass would sit between the g_s that stream still runs and this batch's g_a, on the critical path.
        self.early.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.early):
            x4 = ops.nchw_to_nhwc(x.contiguous(), 4)
        with torch.cuda.stream(stream):
            if self.early_first_layer:                         # (the Manypriors models' encode_batch_begin only)
                enc = self.model.encode_batch_begin(x, input_nhwc=x4, first_layer_on=(self.early, self.deconv2_done))
            else:
                stream.wait_stream(self.early)
                x4.record_stream(stream)
                enc = self.model.encode_batch_begin(x, input_nhwc=x4)
        done = self._fetch_and_decode(enc["transforms_done"]) if self.encoding is not None else None
        self.encoding = (enc, stream, tag, target)
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
        if self.encoding is None:
            return None
        return self._finish(self._fetch_and_decode(None))

    def synchronize(self):
        for s in self.streams + [self.early]:
            s.synchronize()


class ForwardPipeline:
    """Throughput mode of the eval-mode forward (the call the reference's test harness makes once per image,
    src/compression/tests.py:356): `depth` batches in flight on alternating streams, the host's share of each
    (waiting for the index map, which priors were used, the side string's LZMA call) on a worker thread, so that the GPU
    goes from one batch's g_s straight into the next batch's g_a.  submit() returns the dictionary of the batch submitted
    depth - 1 calls earlier: same keys and values as model(x).  A batch whose latents fall outside the tables' range
    is run again after the tables have been widened, like model(x) does; the batches behind it in the pipeline are
    run again too (their rates were estimated with the narrower table)."""

    def __init__(self, model, device, depth=2):
        from concurrent.futures import ThreadPoolExecutor
        if depth < 1:
            raise ValueError("depth must be at least 1")
        self.model = model
        self.device = torch.device(device)
        self.depth = depth
        # one stream per batch in flight (the page-locked readback buffers are per stream: a batch's host part may still
        # be reading its buffer when the batch after next is enqueued) and one worker less than that: with two batches in
        # flight one worker keeps up; the LZMA call, numpy and the event wait all release the interpreter lock
        self.streams = [torch.cuda.Stream(self.device) for _ in range(depth)]
        self.pool = ThreadPoolExecutor(max_workers=max(1, depth - 1))
        self.queue = []              # (input, job, future of its host part, tag, stream), oldest first
        self.count = 0
        for s in self.streams:
            s.wait_stream(torch.cuda.current_stream(self.device))

    def _begin(self, x, tag):
        stream = self.streams[self.count % len(self.streams)]
        self.count += 1
        with torch.cuda.stream(stream):
            job = self.model.forward_begin(x)
        self.queue.append((x, job, self.pool.submit(self.model.forward_host_part, job), tag, stream))

    def _finish_oldest(self):
        x, job, future, tag, stream = self.queue.pop(0)
        with torch.cuda.stream(stream):
            out = self.model.forward_finish(job, future.result())
            if out is None:                                   # out of the tables' range: widen, then everything in flight again
                redo = [(q[0], q[3]) for q in self.queue]
                for q in self.queue:
                    q[2].result()
                self.queue = []
                self.model._widen_tables()
                out = self.model(x)
                for rx, rtag in redo:
                    self._begin(rx, rtag)
        self.result_stream = stream
        return out, tag

    def submit(self, x, tag=None):
        """-> None while the pipeline fills, else (the forward's dictionary, tag) of the oldest batch in flight; its
        tensors are complete on the stream `result_stream`."""
        done = self._finish_oldest() if len(self.queue) >= self.depth else None
        self._begin(x, tag)
        return done

    def flush(self):
        """-> list of (dictionary, tag) of everything still in flight, oldest first."""
        out = []
        while self.queue:
            out.append(self._finish_oldest())
        return out

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        self.pool.shutdown(wait=True)
