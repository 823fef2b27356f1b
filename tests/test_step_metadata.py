"""Per-step metadata producers against sequences recorded from the reference's
own executor code (tests/golden/gen_golden_metadata.py): integer tensors, exact.

CPU tier: the host logic of SlotBatch (padding, rebuild-vs-advance decision, identity table), with
the one-launch advance replaced by the oracle's restatement.  GPU tier: the same scenario through
the HIP kernel, the one-shot bump path through DecodeEngine, and a continuous run on the captured
(batch, bucket) grid checked against every request decoded alone.
"""

import json
import types

import numpy as np
import pytest
import torch

from tests import _golden as G


def _gold():
    return np.load(G.GOLDEN_DIR + "/step_metadata.npz")


# the scripted calls the fixture was recorded for (kept inside the fixture itself)
_S = json.loads(str(_gold()["script"]))
SLOT_SCRIPT, GRAPH_SIZES = _S["slot"], tuple(_S["graph_sizes"])
ONESHOT_LENS, ONESHOT_STEPS = _S["oneshot_lens"], _S["oneshot_steps"]


def _fake_runner(dev):
    from lite_llama_amd.executor import AttentionMetadata, KVPool

    slots, row_len = 5, 16
    pool = KVPool(1, slots * row_len, 1, 8, dev)
    for kv in pool.kv_buffer:
        kv.fill_(1.0)  # so the filler-slot zeroing is observable
    table = torch.zeros(slots, row_len, dtype=torch.int32, device=dev)
    info = AttentionMetadata(kv_buffer=pool.kv_buffer, b_req_tokens_table=table)
    return types.SimpleNamespace(atten_info=info, device=dev, max_seq_len=row_len, b_req_tokens_table=table, pool=pool,
                                 graph_batch_size=lambda n: next((b for b in GRAPH_SIZES if b >= n), n))


def _check_slot_scenario(dev):
    from lite_llama_amd.executor import SlotBatch

    d = _gold()
    runner = _fake_runner(dev)
    sb = SlotBatch(runner)
    info = runner.atten_info
    assert np.array_equal(info.b_req_tokens_table.cpu().numpy(), d["slot.table"])
    assert sb.num_slots == int(d["slot.num_slots"])
    assert runner.pool.max_tokens - runner.pool.used == int(d["slot.free_rows_after_claim"])
    kv = runner.pool.kv_buffer[0]
    assert float(kv[4 * 16:].abs().sum()) == 0.0 and float(kv[: 4 * 16].min()) == 1.0  # only the filler slot is zeroed
    for i, (kind, slots, lens) in enumerate(SLOT_SCRIPT):
        tag = f"slot.{i}"
        if kind == "prefill":
            sb.begin_prefill(slots, lens)
            assert np.array_equal(info.b_start_loc.cpu().numpy().astype(np.int64), d[tag + ".b_start_loc"])
        else:
            padded = sb.begin_decode(slots, lens)
            assert padded == int(d[tag + ".padded"])
            pos = sb.seq_lens.view(-1, 1) - 1
            assert np.array_equal(pos.cpu().numpy().astype(np.int64), d[tag + ".positions"])
            assert info.b_start_loc is None
            # the dtypes the reference hands the kernels (slot_batch.py:211-220 int64; the table gather int32)
            assert info.b_req_idx.dtype == torch.int64 and info.b_seq_len.dtype == torch.int64
            assert info.cur_select_index.dtype == torch.int32
        for key in ("b_req_idx", "b_seq_len", "cur_select_index"):
            assert np.array_equal(getattr(info, key).cpu().numpy().astype(np.int64), d[f"{tag}.{key}"]), (tag, key)
        assert info.max_actual_seq_len == int(d[tag + ".max_actual_seq_len"])
    with pytest.raises(ValueError):
        sb.begin_prefill([0], [17])


def test_slot_batch_host_logic_matches_reference(monkeypatch):
    from lite_llama_amd.executor import slots
    from oracle import oracle as O

    calls = []

    def advance(b_seq_len, b_req_idx, cur_select_index, table):
        calls.append(len(b_seq_len))
        O.slot_advance(b_seq_len, b_req_idx, cur_select_index, table)

    monkeypatch.setattr(slots, "_slot_advance", advance)
    _check_slot_scenario("cpu")
    # steady-state steps of the script (same set, one token further) took the device-side advance
    assert calls == [2, 4, 2, 1]


def test_step_graph_grid_selection():
    from lite_llama_amd.executor.slots import StepGraphs

    g = StepGraphs.__new__(StepGraphs)
    g.batch_sizes, g.seq_len_buckets = (1, 2, 4, 8), (256, 512, 1024)
    assert [g.pad_to(n) for n in (1, 3, 8, 9)] == [1, 4, 8, None]
    assert [g._pick_bucket(n) for n in (1, 256, 257, 1024, 1025)] == [256, 256, 512, 1024, None]


@pytest.mark.gpu
def test_slot_batch_matches_reference_on_device():
    _check_slot_scenario("cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.int32, torch.int64])
def test_slot_advance_kernel(dt):
    from lite_llama_amd.executor import slot_advance
    from oracle import oracle as O

    g = torch.Generator().manual_seed(3)
    slots, row_len, n = 37, 50, 300
    table = torch.randperm(slots * row_len, generator=g).to(torch.int32).view(slots, row_len)
    req = torch.randint(0, slots, (n,), generator=g).to(dt)
    seq = torch.randint(0, row_len, (n,), generator=g).to(dt)
    sel = torch.zeros(n, dtype=torch.int32)
    nxt = torch.randint(0, 1000, (n,), generator=g)
    want_seq, want_sel = seq.clone(), sel.clone()
    O.slot_advance(want_seq, req, want_sel, table)
    tv = table.cuda().t().contiguous().t()  # a strided table view
    seq_d, sel_d = seq.cuda(), sel.cuda()
    pos, ids = torch.zeros(n, 1, dtype=torch.long, device="cuda"), torch.zeros(n, 1, dtype=torch.long, device="cuda")
    slot_advance(seq_d, req.cuda(), sel_d, tv, positions=pos, input_ids=ids, next_tokens=nxt.cuda())
    assert torch.equal(seq_d.cpu(), want_seq) and torch.equal(sel_d.cpu(), want_sel)
    assert torch.equal(pos.view(-1).cpu(), want_seq.long() - 1) and torch.equal(ids.view(-1).cpu(), nxt)
    slot_advance(seq_d[:0], req.cuda()[:0], sel_d[:0], tv)  # empty batch is a no-op


def _tiny_model(quant=None):
    from lite_llama_amd.model import CausalLM, tiny_geometry
    from lite_llama_amd.quantization import QuantConfig

    geo = tiny_geometry(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
                        vocab_size=512)
    return CausalLM(geo).init_synthetic(seed=5, quant=QuantConfig.int4_groupwise(128) if quant else None)


@pytest.mark.gpu
def test_oneshot_metadata_matches_reference_sequence():
    """DecodeEngine's padded-grid prefill + bump/advance produce the tensors of
    prefill_alloc_kv_cache / decode_alloc_kv_cache (model_runner.py:153-218) step for step."""
    from lite_llama_amd.executor import DecodeEngine

    d = _gold()
    eng = DecodeEngine(_tiny_model(), max_batch=len(ONESHOT_LENS), max_seq_len=16)
    info = eng.info
    lp = max(ONESHOT_LENS)
    ids = torch.randint(0, 512, (len(ONESHOT_LENS), lp), device="cuda")
    first = eng.prefill(ids, torch.tensor(ONESHOT_LENS, device="cuda"))

    def same(tag, with_start=False):
        for key in ("b_req_idx", "b_seq_len", "cur_select_index") + (("b_start_loc",) if with_start else ()):
            assert np.array_equal(getattr(info, key).cpu().numpy().astype(np.int64), d[f"{tag}.{key}"]), (tag, key)
        assert np.array_equal(info.b_req_tokens_table.cpu().numpy(), d[tag + ".table"]), tag

    same("oneshot.prefill", with_start=True)
    assert info.max_actual_seq_len == int(d["oneshot.prefill.max_actual_seq_len"])
    seen = []

    def on_step(i):
        same(f"oneshot.{i}")
        assert info.max_actual_seq_len == int(d[f"oneshot.{i}.max_actual_seq_len"])
        seen.append(i)

    eng.decode(first, ONESHOT_STEPS, use_graph=False, on_step=on_step)
    assert seen == list(range(ONESHOT_STEPS))
    assert eng.pool.used == 64 - int(d["oneshot.free_rows"])  # the reference pool of the fixture holds 64 rows


def _solo_logits(model, prompt, fed, max_seq_len):
    """Reference run of one request alone, eager, no slots: logits at every step when fed ``fed``."""
    from lite_llama_amd.executor import SlotRunner

    r = SlotRunner(model, max_request_num=2, max_seq_len=max_seq_len)
    sb = r.enable_slot_kv_cache()
    n = len(prompt)
    sb.begin_prefill([0], [n])
    out = [r.forward(torch.tensor([prompt], device="cuda"), torch.arange(n, device="cuda").view(1, n))[0, n - 1].float()]
    for j, tok in enumerate(fed):
        sb.begin_decode([0], [n + j + 1])
        out.append(r.forward(torch.tensor([[tok]], device="cuda"), sb.seq_lens.view(-1, 1) - 1)[0, -1].float())
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("quant", [None, "int4"])
def test_continuous_run_on_captured_grid_matches_solo_requests(quant):
    """Requests join and leave mid-flight; decode steps run on the (batch, bucket) grid with filler
    rows.  Every request's logits must match the same request decoded alone (the reference's
    tests/engine/test_continuous_batching.py:147-212 properties: co-tenants do not take each
    other's output, survivors are unaffected when a neighbour leaves, graph replay == eager)."""
    from lite_llama_amd.executor import SlotRunner

    torch.manual_seed(0)
    model = _tiny_model(quant)
    max_seq = 64
    runner = SlotRunner(model, max_request_num=5, max_seq_len=max_seq)
    runner.enable_graphs(batch_sizes=(1, 2, 4), seq_len_buckets=(32, 64, 128))
    assert runner._graphs.seq_len_buckets == (32, 64)
    sb = runner.enable_slot_kv_cache()
    prompts = {0: [3, 9, 27, 81, 243], 1: [7, 49, 343], 2: [11, 121, 307, 5]}
    lens, fed, logits, last = {}, {s: [] for s in prompts}, {s: [] for s in prompts}, {}

    def prefill(slot):
        p = prompts[slot]
        sb.begin_prefill([slot], [len(p)])
        lg = runner.forward(torch.tensor([p], device="cuda"), torch.arange(len(p), device="cuda").view(1, -1))
        logits[slot].append(lg[0, len(p) - 1].float())
        last[slot] = int(torch.argmax(lg[0, len(p) - 1]))
        lens[slot] = len(p)

    def decode(running):
        for s in running:
            lens[s] += 1
            fed[s].append(last[s])
        padded = sb.begin_decode(running, [lens[s] for s in running])
        ids = torch.zeros(padded, 1, dtype=torch.long, device="cuda")
        ids[: len(running), 0] = torch.tensor([last[s] for s in running], device="cuda")
        lg = runner.forward(ids, sb.seq_lens.view(-1, 1) - 1)
        assert lg.shape[0] == padded
        for i, s in enumerate(running):
            logits[s].append(lg[i, -1].float())
            last[s] = int(torch.argmax(lg[i, -1]))
        return padded

    prefill(0)
    assert [decode([0]) for _ in range(3)] == [1, 1, 1]
    prefill(1)
    assert [decode([0, 1]) for _ in range(3)] == [2, 2, 2]
    prefill(2)
    assert [decode([0, 1, 2]) for _ in range(4)] == [4, 4, 4, 4]   # padded with the filler slot
    assert [decode([0, 2]) for _ in range(3)] == [2, 2, 2]           # request 1 left
    assert [decode([2]) for _ in range(22)] == [1] * 22              # crosses the 32 -> 64 bucket
    assert set(runner._graphs._graphs) == {(1, 32), (2, 32), (4, 32), (1, 64)}
    for s, p in prompts.items():
        solo = _solo_logits(model, p, fed[s], max_seq)
        assert len(solo) == len(logits[s])
        for j, (a, b) in enumerate(zip(logits[s], solo)):
            torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2, msg=lambda m: f"slot {s} step {j}: {m}")
            top2 = torch.topk(b, 2).values
            if float(top2[0] - top2[1]) > 5e-2:
                assert int(torch.argmax(a)) == int(torch.argmax(b)), (s, j)
