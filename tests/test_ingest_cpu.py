"""CPU tests of the wire/ingest layer: rtl_tcp framing against a loopback server (no GPU, no dongle)."""
import socket
import struct
import threading

import numpy as np
import pytest


class FakeRtlTcp(threading.Thread):
    """Minimal rtl_tcp: sends the 12-byte greeting, then an endless counter-patterned IQ byte stream in small,
    irregular pieces; collects the 5-byte command packets it receives."""

    def __init__(self, stall_after=None, exact=False):
        super().__init__(daemon=True)
        self.exact = exact                       # stop at exactly stall_after stream bytes (else: after the piece crossing it)
        self.srv = socket.socket()
        self.srv.bind(("127.0.0.1", 0))
        self.srv.listen(1)
        self.port = self.srv.getsockname()[1]
        self.commands = b""
        self.stall_after = stall_after
        self.stop = threading.Event()

    @staticmethod
    def stream_bytes(n, start=0):
        i = np.arange(start, start + n, dtype=np.uint64)
        return ((i * 37 + (i >> 8) * 11) & 0xFF).astype(np.uint8)

    def run(self):
        conn, _ = self.srv.accept()
        conn.settimeout(0.01)
        conn.sendall(b"RTL0" + struct.pack(">II", 5, 29))            # magic, tuner type (R820T), gain count
        sent, rng = 0, np.random.default_rng(1)
        try:
            while not self.stop.is_set():
                try:
                    self.commands += conn.recv(64)
                except (socket.timeout, BlockingIOError):
                    pass
                if self.stall_after is not None and sent >= self.stall_after:
                    self.stop.wait(0.05)
                    continue
                n = int(rng.integers(1, 70000))
                if self.exact and self.stall_after is not None:
                    n = min(n, self.stall_after - sent)
                conn.sendall(self.stream_bytes(n, sent).tobytes())
                sent += n
        except (BrokenPipeError, ConnectionResetError, OSError):
            pass
        finally:
            conn.close()
            self.srv.close()


def test_command_packets_match_the_reference(gsmcal_mod):
    ing = gsmcal_mod.ingest
    # set_freq_tcp.m:6-7: byte 1 then uint32(freq) big-endian (MATLAB tcpip byte order)
    assert ing.command_packet(ing.CMD_FREQ, 957400000) == b"\x01" + struct.pack(">I", 957400000)
    assert ing.command_packet(ing.CMD_RATE, 2166667) == b"\x02\x00\x21\x0f\x8b"
    assert ing._matlab_uint32(2166666.6666667) == 2166667 and ing._matlab_uint32(-3) == 0 and ing._matlab_uint32(1e12) == 0xFFFFFFFF


def test_loopback_flush_capture_and_commands(gsmcal_mod):
    ing = gsmcal_mod.ingest
    srv = FakeRtlTcp()
    srv.start()
    d = ing.RtlTcpDongle("127.0.0.1", srv.port, timeout=2.0)
    try:
        n2 = 2 * 50000                                                # 2*num_sample bytes per capture
        d.configure(gain=0, sampling_rate=(1625.0 / 6.0) * 1e3 * 8, freq=957.4e6)      # gsm_sync_demod.m:71-83
        assert d.flush(n2) == n2                                      # :86-89 -- the greeting goes out with the flush
        bufs = [np.zeros(n2, dtype=np.uint8) for _ in range(3)]
        for b in bufs:
            ing.capture_all([d], [memoryview(b)])
        # after discarding 2N bytes (12 of greeting + 2N-12 of samples) the captures continue the stream seamlessly
        off = n2 - ing.RTL_TCP_HEADER_BYTES
        for k, b in enumerate(bufs):
            assert np.array_equal(b, FakeRtlTcp.stream_bytes(n2, off + k * n2)), f"capture {k}"
        # I/Q pairing survives: 12 and 2N are even, so byte 0 of every capture is an I byte
        assert off % 2 == 0
    finally:
        d.close()
        srv.stop.set()
        srv.join(2.0)
    want = (b"\x03" + struct.pack(">I", 0) + b"\x02" + struct.pack(">I", 2166667) + b"\x01" + struct.pack(">I", 957400000))
    assert srv.commands == want                                       # gain mode auto, rate, frequency -- in the driver's order
    # manual gain: mode 1 then the gain value (set_gain_tcp.m:8-11)
    a, b = socket.socketpair()
    try:
        ing.set_gain_tcp(a, 496)
        assert b.recv(16) == b"\x03\x00\x00\x00\x01\x04\x00\x00\x01\xf0"
    finally:
        a.close(); b.close()


def test_short_read_is_retried_then_reported(gsmcal_mod):
    ing = gsmcal_mod.ingest
    srv = FakeRtlTcp(stall_after=150000)                              # the server dries up mid-capture
    srv.start()
    d = ing.RtlTcpDongle("127.0.0.1", srv.port, timeout=0.2)
    try:
        n2 = 100000
        assert d.flush(n2) == n2
        buf = np.zeros(n2, dtype=np.uint8)
        with pytest.raises(IOError):                                  # :99-100 -- the reference loops forever; the mirror gives up loudly
            ing.capture_all([d], [memoryview(buf)], max_tries=2)
    finally:
        d.close()
        srv.stop.set()
        srv.join(2.0)


def test_read_timeout_bounds_the_whole_read_and_pairing_survives_an_odd_short_read(gsmcal_mod):
    """(1) MATLAB's Timeout=1 bounds the whole fread (gsm_sync_demod.m:64) -- a peer that trickles must not
    hold a capture beyond the timeout; (2) a short read of an odd number of bytes must not swap I and Q of every later
    capture."""
    import time
    ing = gsmcal_mod.ingest
    a, b = socket.socketpair()
    try:
        stop = threading.Event()

        def trickle():                                    # one byte every 50 ms: each recv succeeds well inside the timeout
            i = 0
            while not stop.is_set() and i < 200:
                try:
                    b.sendall(bytes([i & 0xFF]))
                except OSError:
                    return
                i += 1
                time.sleep(0.05)

        t = threading.Thread(target=trickle, daemon=True)
        t.start()
        view = memoryview(bytearray(1000))
        t0 = time.monotonic()
        got = ing.read_exact(a, view, timeout=0.4)
        dt = time.monotonic() - t0
        stop.set()
        t.join(2.0)
        assert 0 < got < 1000 and dt < 1.0, (got, dt)     # round 2: every recv re-armed the timeout and this took 50 s
        # the socket leaves read_exact with the CONFIGURED timeout, not the residue of its last recv (the command
        # packets that follow on the same connection would inherit a deadline of microseconds)
        assert a.gettimeout() == pytest.approx(0.4)
    finally:
        a.close(); b.close()

    srv = FakeRtlTcp(stall_after=101, exact=True)         # an ODD number of stream bytes after the greeting, then silence
    srv.start()
    d = ing.RtlTcpDongle("127.0.0.1", srv.port, timeout=0.2)
    try:
        time.sleep(0.3)
        assert d.flush(12) == 12                          # the greeting (kept even so the stream itself starts on an I byte)
        d.parity = 0
        buf = np.zeros(200, dtype=np.uint8)
        got = d.capture_into(memoryview(buf))
        assert got < 200 and got % 2 == 1 and d.parity == 1
        srv.stall_after = None                            # the stream resumes
        buf2 = np.zeros(4000, dtype=np.uint8)
        got2 = d.capture_into(memoryview(buf2))
        assert got2 == 4000 and d.parity == 0
        # one byte was dropped to get back onto an I boundary: the capture starts at an even stream offset
        first = got + 1
        assert first % 2 == 0 and np.array_equal(buf2, FakeRtlTcp.stream_bytes(4000, first))
    finally:
        d.close()
        srv.stop.set()
        srv.join(2.0)
