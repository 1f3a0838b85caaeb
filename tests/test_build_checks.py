"""The build's ISA check for "request now, wait later" inline-asm loads (csrc/dd_convt.hip): pure text logic, runs without hipcc."""
import pytest

from deepdenoiser_amd.build import check_async_loads, parse_resources

REQ, WAIT = "dd_accum_request", "dd_accum_wait"
GOOD = """
_Z4kernPv:
\tv_mov_b32_e32 v1, v0
\tglobal_load_dwordx2 v[10:11], v[2:3], off ; dd_accum_request
\tglobal_load_dwordx2 v[12:13], v[2:3], off ; dd_accum_request
\tv_mfma_f32_16x16x32_bf16 v[20:23], v[4:7], v[14:17], v[20:23]
\tv_add_u32_e32 v9, v8, v1
\ts_waitcnt vmcnt(5) ; dd_accum_wait
\tv_add_f32_e32 v10, v10, v12
\ts_endpgm
"""


def test_clean_isa_passes_and_counts_groups():
    assert check_async_loads(GOOD, REQ, WAIT) == 1
    assert check_async_loads(GOOD + GOOD.replace("_Z4kern", "_Z5kern2"), REQ, WAIT) == 2


@pytest.mark.parametrize("bad", ["\tv_mov_b32_e32 v30, v11", "\tv_pk_add_f16 v[12:13], v[40:41], v[42:43]", "\tscratch_store_dwordx2 off, v[10:11], off offset:16",
                                 "\tv_mfma_f32_16x16x32_bf16 v[20:23], v[4:7], v[10:13], v[20:23]"])
def test_touching_a_pending_register_is_refused(bad):
    text = GOOD.replace("\tv_add_u32_e32 v9, v8, v1", bad)
    with pytest.raises(RuntimeError, match="still in flight"):
        check_async_loads(text, REQ, WAIT)


def test_request_without_wait_is_refused():
    with pytest.raises(RuntimeError, match="never reached"):
        check_async_loads(GOOD.replace("; dd_accum_wait", "").replace("\tv_add_f32_e32 v10, v10, v12\n", ""), REQ, WAIT)


def test_resource_remarks_parse():
    r = parse_resources("remark: x.hip:1:1: Function Name: _Z1kv [-Rpass-analysis]\nremark:     VGPRs: 12 \nremark:     AGPRs: 0\n"
                        "remark:     ScratchSize [bytes/lane]: 8\nremark:     VGPRs Spill: 2\nremark:     Occupancy [waves/SIMD]: 8\nremark:     LDS Size [bytes/block]: 0\n")
    assert r["_Z1kv"]["scratch"] == 8 and r["_Z1kv"]["spill"] == 2 and r["_Z1kv"]["vgprs"] == 12
