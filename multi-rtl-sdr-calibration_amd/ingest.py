"""Wire/ingest layer: rtl_tcp byte stream -> pinned host ring -> asynchronous H2D.

Host-side mirror of the reference's MATLAB socket glue, same packets and same read discipline:
  * command packets  -- set_freq_tcp.m:5-7, set_rate_tcp.m:5-7, set_gain_tcp.m:5-15: one command byte
    (1 = frequency, 2 = sample rate, 3 = gain mode, 4 = tuner gain in tenths of a dB) followed by a uint32 in
    network byte order (MATLAB's tcpip object writes big-endian, which is what rtl_tcp's ntohl expects);
  * flush            -- gsm_sync_demod.m:86-89: the first 2*num_sample bytes after the connection opens are read and
    thrown away; they start with rtl_tcp's own 12-byte greeting ("RTL0", tuner type, gain count), which the reference
    never parses;
  * capture          -- gsm_sync_demod.m:94-104: 2*num_sample interleaved uint8 I,Q bytes per dongle, re-read from
    scratch when any dongle delivered fewer bytes than asked for.
What is new is where the bytes land: recv_into() writes them straight into a pinned slot of a gsmcal_ring (zero copy),
the slot goes to the GPU on the ring's copy stream, and batch k+1 is received and copied while batch k is processed.
"""
from __future__ import annotations

import ctypes as C
import socket
import struct

import numpy as np

RTL_TCP_HEADER_BYTES = 12
CMD_FREQ, CMD_RATE, CMD_GAIN_MODE, CMD_GAIN = 1, 2, 3, 4


def command_packet(cmd, value):
    """One rtl_tcp command: uint8 command + uint32 big-endian parameter (fwrite(tcp_obj, uint32(v), 'uint32'))."""
    return struct.pack(">BI", int(cmd), int(value) & 0xFFFFFFFF)


def set_freq_tcp(sock, freq):
    """set_freq_tcp.m:5-7"""
    sock.sendall(command_packet(CMD_FREQ, _matlab_uint32(freq)))
    return sock


def set_rate_tcp(sock, rate):
    """set_rate_tcp.m:5-7"""
    sock.sendall(command_packet(CMD_RATE, _matlab_uint32(rate)))
    return sock


def set_gain_tcp(sock, gain):
    """set_gain_tcp.m:5-15: gain != 0 -> manual mode (3, 1) then the gain (4, gain); gain == 0 -> automatic (3, 0)."""
    if gain:
        sock.sendall(command_packet(CMD_GAIN_MODE, 1) + command_packet(CMD_GAIN, _matlab_uint32(gain)))
    else:
        sock.sendall(command_packet(CMD_GAIN_MODE, 0))
    return sock


def _matlab_uint32(v):
    """uint32(x) in MATLAB: round half away from zero, saturate to 0 .. 2^32-1."""
    v = float(v)
    r = int(np.sign(v) * np.floor(abs(v) + 0.5))
    return min(max(r, 0), 0xFFFFFFFF)


def read_exact(sock, view, timeout=1.0):
    """fread(tcp_obj, n, 'uint8') with the driver's 1 s timeout (gsm_sync_demod.m:64): fills `view` (a writable
    memoryview) and returns the number of bytes actually received -- fewer than asked for on a timeout or a closed peer.
    MATLAB's Timeout bounds the WHOLE fread, so there is one deadline for the call, not one per recv (a peer trickling a
    byte every 0.9 s would otherwise hold a capture for ever)."""
    import time
    deadline = time.monotonic() + timeout
    got, n = 0, len(view)
    try:
        while got < n:
            left = deadline - time.monotonic()
            if left <= 0.0:
                break
            sock.settimeout(left)
            k = sock.recv_into(view[got:], n - got)
            if k == 0:
                break
            got += k
    except (socket.timeout, BlockingIOError):
        pass
    finally:
        # the socket keeps the configured timeout, not the residue of the last recv: the command packets that follow on the
        # same connection (set_freq_tcp / set_gain_tcp: sendall) must not inherit a deadline of microseconds
        try:
            sock.settimeout(timeout)
        except OSError:
            pass
    return got


class RtlTcpDongle:
    """One rtl_tcp connection, driven like gsm_sync_demod.m:58-90 drives its tcpip object."""

    def __init__(self, host="127.0.0.1", port=1234, timeout=1.0):
        self.sock = socket.create_connection((host, port), timeout=timeout)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self.timeout = timeout
        self.parity = 0                 # 1: an odd number of stream bytes consumed so far (next byte is a Q)
        self.last_flush_short = False

    def configure(self, gain, sampling_rate, freq):
        set_gain_tcp(self.sock, gain)          # gsm_sync_demod.m:71-73
        set_rate_tcp(self.sock, sampling_rate)  # :76-78
        set_freq_tcp(self.sock, freq)           # :81-83

    def flush(self, nbytes):
        """:86-89 -- read and discard 2*num_sample bytes (rtl_tcp's 12-byte greeting is the head of them).  Returns the
        number of bytes discarded; fewer than nbytes means the dongle is not streaming yet (the caller decides)."""
        scratch = memoryview(bytearray(nbytes))
        got = read_exact(self.sock, scratch, self.timeout)
        self.parity ^= got & 1
        self.last_flush_short = got < nbytes
        return got

    def capture_into(self, view):
        """One capture.  The byte stream is I,Q,I,Q,...: a read that stopped after an odd number of bytes would leave the
        NEXT capture starting on a Q byte (I and Q swapped for good), so the pairing is restored first by dropping one
        byte -- the reference never notices this (it only re-reads, gsm_sync_demod.m:97-104)."""
        if self.parity:
            one = memoryview(bytearray(1))
            self.parity ^= read_exact(self.sock, one, self.timeout) & 1
            if self.parity:
                return 0                                   # still mid-pair: report a short capture, the caller retries
        got = read_exact(self.sock, view, self.timeout)
        self.parity ^= got & 1
        return got

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass


def capture_all(dongles, views, max_tries=8):
    """:93-104 -- read one capture from every dongle; if any came up short, read them all again."""
    for _ in range(max_tries):
        counts = [d.capture_into(v) for d, v in zip(dongles, views)]
        if all(c == len(v) for c, v in zip(counts, views)):
            return counts
    raise IOError(f"short reads from rtl_tcp after {max_tries} tries: {counts}")


class Ring:
    """gsmcal_ring of the C ABI: pinned host slots + device twins + a copy stream (see include/gsmcal.h)."""

    def __init__(self, ctx, batch_bytes, slots=2):
        self.ctx, self.bytes, self.slots = ctx, int(batch_bytes), int(slots)
        h = C.c_void_p()
        ctx.check(ctx.lib.gsmcal_ring_create(ctx.h, self.bytes, self.slots, C.byref(h)), "gsmcal_ring_create")
        self.h = h

    def host(self, slot):
        """writable uint8 view of the pinned slot (recv_into target)"""
        p = self.ctx.lib.gsmcal_ring_host(self.h, slot)
        return np.ctypeslib.as_array((C.c_uint8 * self.bytes).from_address(p))

    def submit(self, slot, nbytes=0):
        self.ctx.check(self.ctx.lib.gsmcal_ring_submit(self.h, slot, nbytes), "gsmcal_ring_submit")

    def acquire(self, slot):
        p = self.ctx.lib.gsmcal_ring_acquire(self.h, slot)
        if not p:
            raise RuntimeError("gsmcal_ring_acquire failed")
        return p

    def release(self, slot):
        self.ctx.check(self.ctx.lib.gsmcal_ring_release(self.h, slot), "gsmcal_ring_release")

    def host_ready(self, slot):
        self.ctx.check(self.ctx.lib.gsmcal_ring_host_ready(self.h, slot), "gsmcal_ring_host_ready")

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.gsmcal_ring_destroy(self.h)
            self.h = None
