"""Return of the results on ONE node without a collective: every rank downloads its own records + consensus / allele
bytes over its OWN PCIe link into a POSIX shared-memory segment; the process that merges the junctions and writes the VCF
(`mergeSort` + `vcfOutput`, src/delly.h:232-302, rank 0 here) maps every rank's segment and reads them in place.

Why (CHANGELOG.md 5, "the root funnel"): at ~1 KB of results per junction and tens of millions of junctions per second and
GPU, a gather to one rank pushes every rank's bytes through that rank's single PCIe link; the links of the other GPUs idle.

Segment layout (little endian):
    0   uint64  sequence number of the batch in the segment (odd while the owner is writing: a seqlock)
    8   uint64  n_records
    16  uint64  blob_bytes
    24  uint64  record_bytes (sizeof(dellyhip_result), so that a reader in another language can check its layout)
    64  records (n_records * record_bytes), then at `blob_at` (fixed, 64-byte aligned behind cap_records) the blob

The owner pins its segment with dellyhip_host_register, so dellyhip_batch_fetch writes into it at the PCIe rate.
This module is host plumbing (multiprocessing.shared_memory + numpy views); nothing in it touches the GPU, and the
world-size-2 test on CPU (tests/test_shard_gloo.py) drives exactly this code.
"""
import struct
from multiprocessing import shared_memory

import numpy as np

HEADER = 64


def _name(tag, rank):
    return "dellyhip_%s_r%d" % (tag, rank)


class Segment:
    """one rank's segment; `create=True` for the owner, False for a reader (the merging process)"""

    def __init__(self, tag, rank, cap_records, record_bytes, cap_blob, create):
        self.rank, self.record_bytes = int(rank), int(record_bytes)
        self.cap_records, self.cap_blob = int(cap_records), int(cap_blob)
        self.blob_at = (HEADER + self.cap_records * self.record_bytes + 63) & ~63
        size = self.blob_at + self.cap_blob
        self.owner = bool(create)
        if create:
            try:   # a crashed earlier run may have left the name behind
                old = shared_memory.SharedMemory(name=_name(tag, rank))
                old.close()
                old.unlink()
            except FileNotFoundError:
                pass
            self.shm = shared_memory.SharedMemory(name=_name(tag, rank), create=True, size=size)
            self.shm.buf[:HEADER] = bytes(HEADER)
            struct.pack_into("<Q", self.shm.buf, 24, self.record_bytes)
        else:
            self.shm = shared_memory.SharedMemory(name=_name(tag, rank))
            # (before Python 3.13 every attach registers the segment with this process's resource_tracker, which unlinks it when
            #  the READER exits -- under the owner's feet; only the owner may unlink)
            try:
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self.shm._name, "shared_memory")
            except Exception:   # noqa: BLE001
                pass
            if self.shm.size < size:
                raise ValueError("segment of rank %d is smaller than agreed (%d < %d)" % (rank, self.shm.size, size))
        self.bytes = np.frombuffer(self.shm.buf, dtype=np.uint8)
        self._pinned_by = None

    # ---- owner side ------------------------------------------------------------------------------------------------
    def pin(self, ctx):
        """dellyhip_host_register over the whole segment (owner only, once)"""
        ctx.host_register(self.bytes.ctypes.data, self.bytes.nbytes)
        self._pinned_by = ctx

    def records_view(self):
        return self.bytes[HEADER:HEADER + self.cap_records * self.record_bytes]

    def blob_view(self):
        return self.bytes[self.blob_at:self.blob_at + self.cap_blob]

    def begin(self):
        seq = struct.unpack_from("<Q", self.shm.buf, 0)[0]
        struct.pack_into("<Q", self.shm.buf, 0, seq | 1)          # odd: being written

    def commit(self, n_records, blob_bytes):
        if n_records > self.cap_records or blob_bytes > self.cap_blob:
            raise ValueError("batch does not fit the segment")
        struct.pack_into("<QQ", self.shm.buf, 8, int(n_records), int(blob_bytes))
        seq = struct.unpack_from("<Q", self.shm.buf, 0)[0]
        struct.pack_into("<Q", self.shm.buf, 0, (seq | 1) + 1)    # even: complete

    # ---- reader side -----------------------------------------------------------------------------------------------
    def read(self, dtype, copy=False):
        """-> (sequence number, records as `dtype` view, blob view) of the last committed batch, or None while the owner writes.
        The views are zero-copy and only validated at the moment of the read: the owner's next begin() overwrites them, so a
        reader that consumes them later passes copy=True (or the owner waits for the reader before it calls begin())."""
        seq0, n, nb, rb = struct.unpack_from("<QQQQ", self.shm.buf, 0)
        if seq0 & 1:
            return None
        if rb != np.dtype(dtype).itemsize or rb != self.record_bytes:
            raise ValueError("record layout mismatch: %d bytes in the segment, %d expected" % (rb, np.dtype(dtype).itemsize))
        rec = self.bytes[HEADER:HEADER + n * rb].view(dtype)
        blob = self.bytes[self.blob_at:self.blob_at + nb]
        if copy:
            rec, blob = np.array(rec), np.array(blob)
        if struct.unpack_from("<Q", self.shm.buf, 0)[0] != seq0:
            return None
        return seq0 // 2, rec, blob

    def close(self):
        if self._pinned_by is not None:
            try:
                self._pinned_by.host_unregister(self.bytes.ctypes.data)
            except Exception:   # noqa: BLE001 (the context may be gone already)
                pass
            self._pinned_by = None
        self.bytes = None
        try:
            self.shm.close()
        except BufferError:      # a view handed out by read() / records_view() is still referenced: the mapping stays, the NAME must still go
            pass
        if self.owner:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
