"""Host-side session facade with the reference accelerator's own operator protocol
(SURVEY.md §8f rank 3): load A, load B, start, poll status, read C — the RoCC custom
instructions of chipyard/ntt-test.c:5-64 / NttRocc.scala:98-104,238-241, one coefficient per
instruction, served by the MI355X engine instead of the RTL.

    s = NttRoccSession(4096, 8380417, 283817)
    for i, (x, y) in enumerate(zip(a, b)):
        s.rocc(FUNCT_LOAD_A, i, x); s.rocc(FUNCT_LOAD_B, i, y)
    s.rocc(FUNCT_START)
    while not s.rocc(FUNCT_STATUS) & STATUS_DONE: pass
    c = [s.rocc(FUNCT_READ, i) for i in range(4096)]

mode="cyclic" reproduces what the RTL top level computes (forward -> pointwise -> inverse with
omega = psi^2, no twist: python_poly_mult, test/cocotb_tests/test_ntt_poly_mult.py:38-43);
mode="negacyclic" is nwc_poly_mult (new_reference/cg_ntt.py:78).
"""
from __future__ import annotations

import numpy as np

from . import engine

FUNCT_START, FUNCT_LOAD_A, FUNCT_LOAD_B, FUNCT_READ, FUNCT_STATUS, FUNCT_DEBUG_READ_A, FUNCT_DEBUG_READ_B = range(7)
STATUS_DONE, STATUS_BUSY = 0x1, 0x2
STATUS_FWD_DONE, STATUS_INV_DONE, STATUS_FWD_STARTED, STATUS_INV_STARTED = 0x100, 0x200, 0x400, 0x800
STATE_IDLE, STATE_DONE = 0, 10          # debug_state field, bits [7:4] (ntt-test.c:20,66-69)


class NttRoccSession:
    def __init__(self, n: int, q: int, psi: int, device: int = 0, mode: str = "cyclic", variant: str = "auto", width: int = None):
        if mode not in ("cyclic", "negacyclic"):
            raise ValueError("mode must be 'cyclic' or 'negacyclic'")
        self.plan = engine.get_plan(n, q, psi, device)
        self.n, self.q, self.mode, self.variant = n, q, mode, variant
        self.addr_mask = n - 1                      # load_addr := rs1(addrWidth-1, 0)  (NttRocc.scala:186)
        # load_data := rs2(nttWidth-1, 0) (NttRocc.scala:187; nttWidth = 32 there, :95): the word is TRUNCATED to the
        # coefficient width and stored as is, not reduced; the engine takes any stored word mod q when it is used
        self.width = width if width is not None else (32 if self.plan.elem_bytes == 4 else 64)
        if not 1 <= self.width <= 8 * self.plan.elem_bytes:
            raise ValueError("width must fit the plan's coefficient word")
        self.data_mask = (1 << self.width) - 1
        self._a = np.zeros(n, dtype=self.plan.dtype)
        self._b = np.zeros(n, dtype=self.plan.dtype)
        self._c = np.zeros(n, dtype=self.plan.dtype)
        self._a_ntt = np.zeros(n, dtype=self.plan.dtype)
        self._b_ntt = np.zeros(n, dtype=self.plan.dtype)
        self._status = 0
        self._state = STATE_IDLE

    def rocc(self, funct: int, rs1: int = 0, rs2: int = 0) -> int:
        """One custom instruction; returns rd (0 for instructions without a result)."""
        if funct == FUNCT_LOAD_A:
            self._a[rs1 & self.addr_mask] = rs2 & self.data_mask  # load_data := rs2(width-1, 0)
            return 0
        if funct == FUNCT_LOAD_B:
            self._b[rs1 & self.addr_mask] = rs2 & self.data_mask
            return 0
        if funct == FUNCT_START:
            self._run()
            return 0
        if funct == FUNCT_READ:
            return int(self._c[rs1 & self.addr_mask])
        if funct == FUNCT_STATUS:
            return self._status | (self._state << 4)
        if funct == FUNCT_DEBUG_READ_A:
            return int(self._a_ntt[rs1 & self.addr_mask])
        if funct == FUNCT_DEBUG_READ_B:
            return int(self._b_ntt[rs1 & self.addr_mask])
        return 0                                                  # unknown funct with xd: responds 0 (NttRocc.scala:243-246)

    def _run(self):
        p = self.plan
        self._status = STATUS_BUSY | STATUS_FWD_STARTED
        # debug memories hold the forward transforms of A and B (aNttMem / bNttMem, NttRocc.scala:106-109)
        self._a_ntt = p.ntt_forward(self._a, variant=self.variant)
        self._b_ntt = p.ntt_forward(self._b, variant=self.variant)
        self._status |= STATUS_FWD_DONE | STATUS_INV_STARTED
        if self.mode == "cyclic":
            self._c = p.cyclic_poly_mult(self._a, self._b, variant=self.variant)
        else:
            self._c = p.poly_mult(self._a, self._b, variant=self.variant)
        self._status = STATUS_DONE | STATUS_FWD_DONE | STATUS_INV_DONE | STATUS_FWD_STARTED | STATUS_INV_STARTED
        self._state = STATE_DONE

    # convenience mirroring chipyard/ntt-test.c:main (:91-172)
    def multiply(self, a, b):
        for i in range(self.n):
            self.rocc(FUNCT_LOAD_A, i, int(a[i]) if i < len(a) else 0)
            self.rocc(FUNCT_LOAD_B, i, int(b[i]) if i < len(b) else 0)
        self.rocc(FUNCT_START)
        if not self.rocc(FUNCT_STATUS) & STATUS_DONE:
            raise RuntimeError("accelerator did not finish")
        return [self.rocc(FUNCT_READ, i) for i in range(self.n)]
