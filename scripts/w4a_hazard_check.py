#!/usr/bin/env python3
"""Race / hazard checker for the generated four-wave GEMM bodies (scripts/gen_w4a.py) -- runs on the CPU, no GPU, no assembler.

The asm kernel's correctness hangs on COUNTED waits: `s_waitcnt vmcnt(n)` / `lgkmcnt(n)` with n derived from the order in which the generator issued
its loads, and on two workgroup barriers per K-step that hand an LDS stage from the LDS-DMA to the fragment reads and back.  A wait that is one too weak
reads a register (or an LDS stage) a load has not filled yet -- on the hardware that is a wrong result some of the time.  This script executes a body
symbolically, instruction by instruction, through its real control flow (scalar counters interpreted: K-step pairs, tiles) and PROVES, for the path
taken, under the weakest assumption the ISA allows (a memory operation has completed only when a wait or the counter's width says so):

  R1  no instruction reads or overwrites a VGPR that an outstanding load (global -> VGPR, LDS -> VGPR) still has to write;
  R2  a fragment read (ds_read) of LDS stage s happens only after every piece of that stage's K-step has been waited for (vmcnt) AND a barrier
      has been passed since (the other waves' pieces), and before the next K-step's DMA starts writing it;
  R3  the LDS-DMA starts refilling stage s only after every fragment read of the previous K-step has returned (lgkmcnt) AND a barrier has been
      passed since (the other waves' reads);
  R4  every K-step's DMA is complete (ni + 8 pieces into each of the A and B halves) when it is published, every counted wait fits its counter;
  R5  at the end nothing is outstanding that the kernel has not waited for (the final vmcnt(0) lgkmcnt(0));
  R6-R10  the fixed-latency hazards gfx940-class hardware does NOT interlock, counted in wait states (one per instruction, n + 1 per `s_nop n`):
      an MFMA's accumulator read by v_accvgpr_read (>= 11: covers matrix operations of up to 8 passes), a VALU result read by a DPP instruction (>= 2),
      a transcendental's result read by another VALU instruction (>= 1), m0 written by the SALU and used by an LDS-DMA (>= 1), an SGPR written by
      v_readfirstlane and used by a buffer instruction (>= 5);
  R13  (round 6, from /opt/skills/guides/cdna_hip_programming.md section 5.7) the data registers of a wide store (buffer_store_dwordx3 / x4) are not written within two wait states
      behind it (the store reads them late; the generator pads with `s_nop 1`), and R6 asks for the guide's 12 states behind an MFMA instead of the ISA table's 11.

Loads and stores retire in order per counter (gfx950: ONE in-order vmcnt for global loads, LDS-DMA and stores; LDS reads return in order), which is what
the generator assumes; a counter of w bits cannot hold more than 2^w - 1 operations, so the oldest of more than that has completed.

    python scripts/w4a_hazard_check.py            (all bodies x a few (K, tiles) shapes; exit code 1 on the first violation)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_w4a as G  # noqa: E402


class Violation(Exception):
    pass


REG_V = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
IMM = re.compile(r"(?<![\w\[:])(-?\d+|0x[0-9a-fA-F]+)\b")


def vregs(tok):
    out = []
    for m in REG_V.finditer(tok):
        if m.group(1) is not None:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


class Sim:
    VM_MAX, LGKM_MAX = 63, 15

    def __init__(self, lines, ni, K, ntiles, name=""):
        self.lines, self.ni, self.K, self.ntiles, self.name = lines, ni, K, ntiles, name
        self.labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
        self.s = {}                      # scalar registers with known values
        self.scc = 0
        self.vm = []                     # VMEM operations in issue order: dict(id, done, kind, ...)
        self.lg = []                     # LGKM operations (ds_read, s_load)
        self.pend = {}                   # VGPR -> the operation that still has to write it
        # LDS regions: (operand, stage) -> state
        self.reg = {(o, s_): dict(state="EMPTY", dma=[], reads=[], drained=0) for o in "AB" for s_ in (0, 1)}
        self.m0 = None                   # (operand, stage) the next LDS-DMA writes
        self.n_instr = 0
        self.stats = dict(mfma=0, ksteps=0, tiles=0)
        self.ws = 0                      # wait-state clock: instructions issued so far (s_nop n counts n + 1)
        self.acc_w, self.valu_w, self.trans_w, self.rfl_w = {}, {}, {}, {}   # register -> wait-state clock right after the instruction that wrote it
        self.m0_w = -100
        self.store_w = {}                # VGPR -> wait-state clock right behind the wide store that reads it as data (R13)

    # ------------------------------------------------------------------ helpers
    def fail(self, i, msg):
        raise Violation(f"{self.name} K={self.K} tiles={self.ntiles}: line {i}: {self.lines[i]!r}: {msg}")

    def val(self, tok):
        tok = tok.strip()
        if tok.startswith("%"):
            name = G.OPS[int(tok[1:])]
            return self.ntiles if name == "ntiles" else 0
        m = re.fullmatch(r"s(\d+)", tok)
        if m:
            return self.s.get(int(m.group(1)))
        if tok == "m0":
            return None
        try:
            return int(tok, 0)
        except ValueError:
            return None

    def issue(self, queue, op, width):
        op["done"] = False
        queue.append(op)
        live = [o for o in queue if not o["done"]]
        while len(live) > width:         # the counter cannot hold more: the oldest has retired
            self.retire(live.pop(0))

    def retire(self, op):
        if op["done"]:
            return
        op["done"] = True
        for r in op.get("dst", []):
            if self.pend.get(r) is op:
                del self.pend[r]

    def wait(self, queue, n):
        live = [o for o in queue if not o["done"]]
        for o in live[: max(0, len(live) - n)]:
            self.retire(o)

    def read_v(self, i, regs):
        for r in regs:
            if r in self.pend:
                self.fail(i, f"R1: reads v{r} while {self.pend[r]['what']} (line {self.pend[r]['line']}) has not been waited for")

    def write_v(self, i, regs):
        for r in regs:
            if r in self.store_w and self.ws - self.store_w[r] < 2:
                self.fail(i, f"R13: v{r} is written {self.ws - self.store_w[r]} wait states behind the dwordx4 store that reads it as data (2 required)")
            if r in self.pend:
                self.fail(i, f"R1: overwrites v{r} while {self.pend[r]['what']} (line {self.pend[r]['line']}) is still to write it")

    # ------------------------------------------------------------------ LDS protocol
    def lds_read(self, i, operand, stage, op):
        rg = self.reg[(operand, stage)]
        if rg["state"] != "READY":
            self.fail(i, f"R2: fragment read of {operand} stage {stage} in state {rg['state']} (not published by a barrier behind a vmcnt wait)")
        rg["reads"].append(op)

    def lds_dma(self, i, operand, stage, op):
        rg = self.reg[(operand, stage)]
        if rg["state"] == "READY":
            if rg["drained"] != len(rg["reads"]) or any(not o["done"] for o in rg["reads"]):
                self.fail(i, f"R3: LDS-DMA into {operand} stage {stage} while fragment reads of it are outstanding or no barrier has been passed since they returned "
                             f"({rg['drained']} of {len(rg['reads'])} reads covered)")
            rg.update(state="FILLING", dma=[], reads=[], drained=0)
        elif rg["state"] == "EMPTY":
            rg.update(state="FILLING", dma=[])
        rg["dma"].append(op)
        need = self.ni if operand == "A" else 8
        if len(rg["dma"]) > need:
            self.fail(i, f"R4: more than {need} DMA pieces into {operand} stage {stage} before it was published")

    def barrier(self, i):
        for (operand, stage), rg in self.reg.items():
            need = self.ni if operand == "A" else 8
            if rg["state"] == "FILLING" and len(rg["dma"]) == need and all(o["done"] for o in rg["dma"]):
                rg.update(state="READY", reads=[], drained=0)
                if operand == "A":
                    self.stats["ksteps"] += 1
            elif rg["state"] == "READY" and all(o["done"] for o in rg["reads"]):
                rg["drained"] = len(rg["reads"])

    # ------------------------------------------------------------------ one instruction
    def gap(self, i, table, regs, need, rule):
        for r in regs:
            if r in table and self.ws - table[r] < need:
                self.fail(i, f"{rule}: {self.ws - table[r]} wait states since the producer of register {r}, {need} required")

    def step(self, i):
        nxt = self.step_(i)
        ln = self.lines[i]
        if not ln.endswith(":"):
            self.ws += (int(ln.split()[1], 0) + 1) if ln.startswith("s_nop") else 1
        return nxt

    def step_(self, i):
        ln = self.lines[i]
        self.n_instr += 1
        if ln.endswith(":"):
            return i + 1
        op, _, rest = ln.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        if op.startswith("buffer_"):            # R10: SGPRs a buffer instruction uses (descriptor, scalar offset) against v_readfirstlane
            used = []
            for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", rest):
                used += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
            self.gap(i, self.rfl_w, used, 5, "R10")
            if rest.rstrip().endswith(" lds") and self.ws - self.m0_w < 1:
                self.fail(i, "R9: LDS-DMA in the wait state right behind the SALU write of m0")
        # ---- control flow / scalar
        if op == "s_branch":
            return self.labels[args[0]]
        if op in ("s_cbranch_scc0", "s_cbranch_scc1"):
            take = (self.scc == 0) if op.endswith("0") else (self.scc == 1)
            return self.labels[args[0]] if take else i + 1
        if op in ("s_cmp_eq_u32", "s_cmp_lt_u32"):
            a, b = self.val(args[0]), self.val(args[1])
            if a is None or b is None:
                self.scc = 0               # (only the descriptor selects compare unknown values: either outcome is a valid path)
            else:
                self.scc = int(a == b) if op == "s_cmp_eq_u32" else int(a < b)
            return i + 1
        if op.startswith("s_load_dword"):
            m = re.match(r"s\[(\d+):(\d+)\]|s(\d+)", args[0])
            lo, hi = (int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)), int(m.group(3)))
            for r in range(lo, hi + 1):
                self.s[r] = 0
            if lo <= G.S_Kd <= hi:
                self.s[G.S_Kd] = self.K
            self.issue(self.lg, dict(line=i, what="s_load", dst=[]), self.LGKM_MAX)
            return i + 1
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                if int(m.group(1)) > self.VM_MAX:
                    self.fail(i, "R4: vmcnt beyond 6 bits")
                self.wait(self.vm, int(m.group(1)))
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                if int(m.group(1)) > self.LGKM_MAX:
                    self.fail(i, "R4: lgkmcnt beyond 4 bits")
                self.wait(self.lg, int(m.group(1)))
            return i + 1
        if op == "s_barrier":
            self.barrier(i)
            return i + 1
        if op.startswith("s_"):
            if op in ("s_nop",):
                return i + 1
            dst = args[0]
            if dst == "m0":
                self.m0_w = self.ws + 1
                base = int(re.fullmatch(r"s(\d+)", args[1]).group(1))
                const = int(args[2], 0)
                self.m0 = ("A" if base == G.S_M0A else "B", 1 if const >= 32768 else 0)
                return i + 1
            m = re.fullmatch(r"s(\d+)", dst)
            if m:
                d = int(m.group(1))
                a = self.val(args[1]) if len(args) > 1 else None
                b = self.val(args[2]) if len(args) > 2 else None
                v = None
                if op == "s_mov_b32":
                    v = a
                elif a is not None and b is not None:
                    v = {"s_add_u32": a + b, "s_sub_u32": a - b, "s_lshr_b32": a >> b, "s_lshl_b32": a << b, "s_mul_i32": a * b,
                         "s_and_b32": a & b, "s_or_b32": a | b, "s_max_i32": max(a, b)}.get(op)
                if v is not None:
                    v &= 0xFFFFFFFF
                self.s[d] = v
            return i + 1
        # ---- LDS reads
        if op.startswith("ds_read"):
            dst = vregs(args[0])
            addr = vregs(args[1].split()[0])
            self.read_v(i, addr)
            self.write_v(i, dst)
            m = re.search(r"offset:(\d+)", rest)
            off = int(m.group(1)) if m else 0
            o = dict(line=i, what="a fragment read (ds_read)", dst=dst)
            if addr[0] in G.V_AA:
                self.lds_read(i, "A", 1 if off >= 32768 else 0, o)
            elif addr[0] in G.V_AB:
                self.lds_read(i, "B", 1 if off >= 32768 else 0, o)
            self.issue(self.lg, o, self.LGKM_MAX)
            for r in dst:
                self.pend[r] = o
            return i + 1
        # ---- VMEM
        if op.startswith("buffer_load"):
            if rest.rstrip().endswith(" lds"):
                self.read_v(i, vregs(args[0]))
                if self.m0 is None:
                    self.fail(i, "LDS-DMA without m0")
                o = dict(line=i, what="an LDS-DMA piece", dst=[])
                self.issue(self.vm, o, self.VM_MAX)
                self.lds_dma(i, self.m0[0], self.m0[1], o)
                return i + 1
            dst = vregs(args[0])
            self.read_v(i, vregs(args[1]))
            self.write_v(i, dst)
            o = dict(line=i, what="a global load", dst=dst)
            self.issue(self.vm, o, self.VM_MAX)
            for r in dst:
                self.pend[r] = o
            return i + 1
        if op.startswith("buffer_store"):
            if op in ("buffer_store_dwordx4", "buffer_store_dwordx3"):
                for r in vregs(args[0]):
                    self.store_w[r] = self.ws + 1
            self.read_v(i, vregs(args[0]) + vregs(args[1]))
            self.issue(self.vm, dict(line=i, what="a store", dst=[]), self.VM_MAX)
            return i + 1
        # ---- matrix / vector ALU: first operand is the destination, the rest are sources
        if op.startswith("v_mfma"):
            self.read_v(i, vregs(args[1]) + vregs(args[2]))
            self.stats["mfma"] += 1
            m = re.fullmatch(r"a\[(\d+):(\d+)\]", args[0])
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                self.acc_w[r] = self.ws + 1
            return i + 1
        if op.startswith("v_"):
            srcs = []
            for a in args[1:]:
                srcs += vregs(a)
            if op.startswith("v_cmp"):
                srcs += vregs(args[0])
                dst = []
            elif op == "v_readfirstlane_b32":
                dst = []
            else:
                dst = vregs(args[0])
            if "_dpp" in op or op in ("v_cndmask_b32",):
                srcs += dst                 # (bank-masked DPP moves / selects keep part of the old value)
            self.read_v(i, srcs)
            self.write_v(i, dst)
            if op == "v_accvgpr_read_b32":
                self.gap(i, self.acc_w, [int(args[1][1:])], 12, "R6")
            if "_dpp" in op:
                self.gap(i, self.valu_w, vregs(args[1].split()[0]), 2, "R7")
            trans = op in ("v_rcp_f32", "v_exp_f32")
            if not trans:
                self.gap(i, self.trans_w, srcs, 1, "R8")
            for r in dst:
                self.valu_w[r] = self.ws + 1
                if trans:
                    self.trans_w[r] = self.ws + 1
                else:
                    self.trans_w.pop(r, None)
            if op == "v_readfirstlane_b32":
                self.rfl_w[int(args[0][1:])] = self.ws + 1
            return i + 1
        self.fail(i, "instruction the checker does not know")

    def run(self):
        i, n = 0, len(self.lines)
        while i < n:
            i = self.step(i)
            if self.n_instr > 5_000_000:
                raise Violation(f"{self.name}: does not terminate")
        if any(not o["done"] for o in self.vm + self.lg) or self.pend:
            raise Violation(f"{self.name} K={self.K} tiles={self.ntiles}: R5: operations outstanding at the end of the kernel")
        pairs = self.K // 128
        want_k = 2 * pairs * self.ntiles + 1                      # (the last tile's last pair prefetches two K-steps past the end through null descriptors;
                                                                  #  the first of them is still published by the last step's barrier, the second only waited for)
        want_mfma = 2 * pairs * self.ntiles * 2 * self.ni * 8
        if self.stats["mfma"] != want_mfma:
            raise Violation(f"{self.name} K={self.K} tiles={self.ntiles}: executed {self.stats['mfma']} MFMAs, expected {want_mfma}")
        if self.stats["ksteps"] != want_k:
            raise Violation(f"{self.name} K={self.K} tiles={self.ntiles}: published {self.stats['ksteps']} K-steps, expected {want_k}")
        return self.stats


def bodies():
    for ni, bkm in ((8, False), (8, True), (7, False), (7, True)):
        for epi in ("plain", "resid", "mulaux") + (() if bkm else ("dropres", "gelu", "gelud")):
            yield ni, bkm, epi


def check_all(shapes=((256, 1), (256, 3), (384, 2), (768, 2)), verbose=False, opts=()):
    n = 0
    for ni, bkm, epi in bodies():
        lines = G.Gen(bkm, epi, ni, opts).body()
        for K, ntiles in shapes:
            st = Sim(lines, ni, K, ntiles, name=f"ni={ni} {'KM' if bkm else 'KC'} {epi}").run()
            n += 1
            if verbose:
                print(f"ni={ni} {'KM' if bkm else 'KC'} {epi:8s} K={K:4d} tiles={ntiles}: {st['mfma']} MFMAs, {st['ksteps']} K-steps published: ok", flush=True)
    return n


if __name__ == "__main__":
    try:
        n = check_all(verbose=True)
    except Violation as e:
        print("VIOLATION:", e)
        sys.exit(1)
    print(f"{n} (body, shape) runs: every counted wait and both barriers hold")
