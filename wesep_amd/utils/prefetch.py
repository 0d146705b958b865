"""Datapipe hand-off: pinned-memory, non-blocking H2D copies on a dedicated HIP
stream, one batch ahead of the step that consumes it.

The reference moves every batch with blocking `.to(device)` calls inside the step (`executor.py:83-86`); here the
CPU datapipe (`wesep/dataset/*`, unchanged: north_star keeps simulation on the CPU) hands a collated batch dict
(`tse_collate_fn`, `dataset.py:206-264`) to `DevicePrefetcher`, which

  * pins the float tensors of batch i+1 (if the DataLoader did not already: `pin_memory=True`),
  * issues their H2D copies on a copy stream while step i computes,
  * makes the consumer's stream wait on the copy's event before batch i+1 is used, and keeps the pinned
    source alive until then (`record_stream`).

Tensors are converted to float32 like `executor.py:83-85`; integer tensors (speaker labels) keep their dtype;
non-tensor entries (utterance keys) pass through.  On a CPU device (gloo tests) it is a plain pass-through."""
import torch


class DevicePrefetcher:
    def __init__(self, loader, device, float_keys=("wav_mix", "wav_targets", "spk_embeds")):
        self.loader, self.device = loader, torch.device(device)
        self.float_keys = set(float_keys)
        self.cuda = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        """Start the copies of one batch; returns (device batch, event or None)."""
        if not self.cuda:
            return {k: (v.float() if k in self.float_keys and isinstance(v, torch.Tensor) else v)
                    for k, v in batch.items()}, None
        out = {}
        with torch.cuda.stream(self.copy_stream):
            for k, v in batch.items():
                if isinstance(v, torch.Tensor):
                    if k in self.float_keys:
                        v = v.float()
                    if not v.is_pinned():
                        v = v.pin_memory()
                    out[k] = v.to(self.device, non_blocking=True)
                else:
                    out[k] = v
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._stage(next(it))          # batch i+1 copies while the caller computes on batch i
            except StopIteration:
                nxt = None
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for v in cur.values():               # allocated on the copy stream, consumed on the current one
                    if isinstance(v, torch.Tensor):
                        v.record_stream(torch.cuda.current_stream(self.device))
            yield cur
