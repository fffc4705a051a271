"""The shipped device code is free of a gfx950 store-data hazard the compiler's hazard recognizer does not cover
(tools/check_store_hazard.py; measured in tools/ubench/store_data_hazard.hip): a VALU write of a data register of a
VMEM store of more than 64 bits within two instructions of the store reaches memory in some lanes."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
chk = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(chk)


def test_scanner_flags_the_pattern_and_only_the_pattern():
    bad = ["_ZN6kernelE:", "\tbuffer_store_dwordx4 v[4:7], v63, s[36:39], s0 offen", "\tv_cndmask_b32_e64 v4, 0, 1, s[28:29]"]
    late = ["_ZN6kernelE:", "\tbuffer_store_dwordx4 v[4:7], v63, s[36:39], 0 offen", "\ts_nop 0", "\tv_mov_b32_e32 v7, 0"]
    ok = ["_ZN6kernelE:", "\tbuffer_store_dwordx4 v[4:7], v63, s[36:39], s0 offen", "\ts_nop 0", "\ts_nop 0", "\tv_mov_b32_e32 v4, 0",
          "\tglobal_store_dwordx4 v[0:1], v[8:11], off", "\tv_add_u32_e32 v12, 1, v8", "\tv_cmp_gt_i32_e32 vcc, v8, v9",
          "\tbuffer_store_dwordx2 v[4:5], v63, s[36:39], 0 offen", "\tv_mov_b32_e32 v4, 0"]
    assert len(chk.scan(bad, "t")) == 1
    assert len(chk.scan(late, "t")) == 1            # one instruction in between is not enough (immediate soffset)
    assert chk.scan(ok, "t") == []
    glob = ["_ZN6kernelE:", "\tglobal_store_dwordx4 v[0:1], v[8:11], off", "\tv_mov_b32_e32 v9, 0"]
    assert len(chk.scan(glob, "t")) == 1


def test_scanner_counts_wait_states_and_follows_branches():
    # s_nop 1 is two wait states: the write behind it is safe
    nop1 = ["_ZN6kernelE:", "\tbuffer_store_dwordx4 v[4:7], v63, s[36:39], s0 offen", "\ts_nop 1", "\tv_mov_b32_e32 v4, 0"]
    assert chk.scan(nop1, "t") == []
    # a branch inside the window: the scan goes on at its target (objdump listings carry the addresses)
    def listing(branch):
        return ["0000000000001000 <_Zkern>:",
                "\tbuffer_store_dwordx4 v[4:7], v1, s[0:3], s5 offen  // 000000001000: E07C1000 05000401",
                "\t%s  // 000000001008: BF820002" % branch,
                "\tv_mov_b32_e32 v9, 0   // 00000000100C: 7E120280",
                "\tv_mov_b32_e32 v9, 0   // 000000001010: 7E120280",
                "\tv_mov_b32_e32 v4, 0   // 000000001014: 7E080280"]
    assert len(chk.scan(listing("s_branch 2"), "t")) == 1
    assert len(chk.scan(listing("s_cbranch_scc1 2"), "t")) == 1
    assert chk.scan(listing("s_nop 1"), "t") == []
    # without addresses the successor is unknown: reported
    unknown = ["_ZN6kernelE:", "\tbuffer_store_dwordx4 v[4:7], v63, s[36:39], s0 offen", "\ts_branch 12", "\tv_mov_b32_e32 v8, 0"]
    assert len(chk.scan(unknown, "t")) == 1


def test_the_shipped_library_has_no_store_data_hazard():
    so = os.path.join(ROOT, "unflow_amd", "libunflow_hip.so")
    assert os.path.exists(so), "build the library first (__graft_entry__.build())"
    found, n_stores = chk.check([so])
    assert n_stores > 20, "the disassembly found no wide stores: the scan did not see the device code"
    assert found == [], "\n".join(found)
