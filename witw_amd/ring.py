"""Page-locked, shared staging memory between DataLoader workers and the GPU copy stream.

torch's DataLoader moves a worker's batch three times on the host before the GPU sees it: the worker's tensors are copied into
shared memory when the batch is pickled, the parent maps that (fresh) shared-memory file, and its pinning thread copies the batch
once more into page-locked memory -- one thread, about 3.5 GB/s, which caps the cvig_fov data path at ~3.7 k pairs/s whatever
the workers do. Here the parent owns ONE block of shared memory, page-locks it once (hipHostRegister) and hands slots of it to the
workers through a queue: a worker builds its batch block directly in a slot (collate_packed(ring=...)) and sends back a few
integers; the parent starts the host-to-device DMA straight from the slot and returns the slot when the copy's event has fired.
The reference has no counterpart (its DataLoader hands float tensors to `.to(device)`, model/cvig_fov.py:402, :440-442)."""
import multiprocessing as mp

import torch


class PinnedRing(object):
    """slots: at least num_workers x prefetch_factor + 2. A DataLoader hands batches out IN ORDER, so a worker that runs ahead
    holds slots for batches the parent cannot take yet; with fewer slots than batches the workers may have in flight, the worker
    that owes the NEXT batch can starve for a slot (deadlock). acquire() raises after `timeout` seconds instead of hanging."""

    def __init__(self, slots, slot_bytes):
        self.slots, self.slot_bytes = int(slots), int(slot_bytes)
        self.mem = torch.empty((self.slots, self.slot_bytes), dtype=torch.uint8).share_memory_()
        self.registered = False
        if torch.cuda.is_available():
            rc = torch.cuda.cudart().cudaHostRegister(self.mem.data_ptr(), self.slots * self.slot_bytes, 0)
            if int(rc) != 0:
                raise RuntimeError('hipHostRegister of %d MB failed (%s)' % (self.slots * self.slot_bytes >> 20, rc))
            self.registered = True
        self.free = mp.get_context('fork').Queue()      # workers are forked: they inherit the mapping and the queue
        for i in range(self.slots):
            self.free.put(i)
        self._reaper, self._todo = None, None           # parent: thread returning slots behind their copy events

    # ---- worker side
    def acquire(self, timeout=600.0):
        return self.free.get(timeout=timeout)

    def allocator(self, slot):
        """-> alloc(nbytes): consecutive 16-byte aligned uint8 views of the slot (numpy), None once the slot is full."""
        row = self.mem[slot].numpy()
        state = {'off': 0}

        def alloc(nbytes):
            o = state['off']
            if o + nbytes > self.slot_bytes:
                return None
            state['off'] = (o + nbytes + 15) // 16 * 16
            return o, row[o:o + nbytes]
        return alloc

    # ---- parent side
    def view(self, slot, off, nbytes):
        return self.mem[slot, off:off + nbytes]

    def release_after(self, slot, event):
        """Return `slot` to the workers once `event` (recorded behind the host-to-device copies out of it) has fired. A small
        reaper thread waits on the events: the parent's main thread may be blocked in the DataLoader waiting for exactly the
        batch whose worker is waiting for this slot."""
        import queue
        import threading
        if self._reaper is None:
            self._todo = queue.Queue()

            def run():
                while True:
                    item = self._todo.get()
                    if item is None:
                        return
                    ev, sl = item
                    ev.synchronize()
                    self.free.put(sl)
                    self._todo.task_done()
            self._reaper = threading.Thread(target=run, name='witw-ring-reaper', daemon=True)
            self._reaper.start()
        self._todo.put((event, slot))

    def reap(self, wait=False):
        """wait=True: block until every slot handed to release_after is back in the free queue."""
        if wait and self._reaper is not None:
            self._todo.join()

    def release(self, slot):
        self.free.put(slot)

    def close(self):
        self.reap(wait=True)
        if self._reaper is not None:
            self._todo.put(None)
            self._reaper.join(5)
            self._reaper = None
        if self.registered:
            torch.cuda.cudart().cudaHostUnregister(self.mem.data_ptr())
            self.registered = False
