"""GPU twin of tests/test_pack_guard_cpu.py: the kernels take the decisions their CPU model takes, on inputs built to sit
on the edge of the guards.
 * k_pack<.., SPEC> through c1_pack_spec_tap_device on adversarial coefficients (the reference's, moved 0.95 of the bound
   towards the nearest truncation boundary; bounds from realistic to 1000 x looser): mantissas == model, redo /
   reallocation / re-analysis lists == the model's flags -- so what the CPU test proves of the model (every unit whose
   bytes would differ from the reference's is flagged) holds for the kernel;
 * the scale-factor guard of k_analysis_spec through c1_spec_stages_device on twelve signal classes: indices and flag ==
   model on the kernel's own coefficients and bounds."""
import numpy as np
import pytest

import oracle_lib as O
import pack_model_lib as P
import spec_model_lib as M
from test_pack_guard_cpu import adversarial, material, reference_units
from test_spec_bound import signals

pytestmark = pytest.mark.gpu
AMOUNTS = [20, 28, 32, 36, 40, 44, 48, 52]


@pytest.fixture(scope='module')
def ctx():
    import carta1_amd as c1
    c = c1.Context(0)
    yield c
    c.close()


def alloc_record(nbfu, wl):
    rec = np.zeros(32, dtype=np.uint8)
    for b in range(52):
        w = int(wl[b]) if b < nbfu else 0
        rec[b >> 1] |= w << (4 * (b & 1))
    rec[31] |= AMOUNTS.index(nbfu) << 4                       # dword 7 bits 28..30
    return rec


def run_tap(ctx, coefs, eps4, side, alloc):
    import torch
    n = coefs.shape[0]
    d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (coefs, eps4, side, alloc)]
    out = torch.zeros(n * 212, dtype=torch.uint8, device='cuda')
    lists = torch.zeros(8 + 3 * n, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    ctx.pack_spec_tap_device(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), n, out.data_ptr(), lists.data_ptr())
    ctx.synchronize()
    ls = lists.cpu().numpy().astype(np.int64)
    sets = [set(ls[8 + k * n:8 + k * n + ls[k]].tolist()) for k in range(3)]
    assert all(len(sets[k]) == ls[k] for k in range(3))       # no unit listed twice
    return out.cpu().numpy().reshape(n, 212), sets


@pytest.mark.parametrize('scale', [1.0, 30.0, 1000.0], ids=['bound', 'bound_x30', 'bound_x1000'])
def test_speculative_quantizer_equals_its_model_on_adversarial_coefficients(ctx, scale):
    rng = np.random.default_rng(5)
    coefs, eps4, side, alloc, want_q, want_doubt, want_open, exact_flag, fields = [], [], [], [], [], [], [], [], []
    for name, pcm in material():
        _, eps_all, _ = M.run(pcm)
        for f, (ref, n, wl, sfi, _) in enumerate(reference_units(pcm)):
            if f == 0:
                continue
            kind = (f + len(coefs)) % 3                        # 0: speculative unit; 1: with an open scale factor; 2: exact coefficients (bounds of zero)
            eps = (eps_all[f] * scale).astype(np.float32)
            slots = P.to_slots(ref)
            F = adversarial(slots, eps, sfi, wl, n, 0.95, rng) if kind != 2 else slots
            if kind == 2:
                eps = np.zeros(3, dtype=np.float32)
            q, doubtful, _ = P.quantize(F, eps, sfi, wl, n)
            coefs.append(F)                                    # all long: slot order == coefficient order
            eps4.append(np.concatenate([eps, np.array([1 if kind == 1 else 0], dtype=np.uint32).view(np.float32)]))
            s = np.zeros(64, dtype=np.uint8)
            s[:52] = sfi
            side.append(s)
            alloc.append(alloc_record(n, wl))
            want_q.append(q)
            want_doubt.append(doubtful)
            want_open.append(kind == 1)
            exact_flag.append(kind == 2)
            fields.append((n, wl, sfi))
    units, (redo, realloc, reana) = run_tap(ctx, np.array(coefs), np.array(eps4), np.array(side), np.array(alloc))
    n_units = len(coefs)
    assert any(want_doubt) and not all(want_doubt)
    for u in range(n_units):
        fld = O.unpack_unit(units[u])
        n, wl, sfi = fields[u]
        assert fld.nbfu == n and list(fld.wl[:n]) == list(wl[:n]) and list(fld.sfi[:n]) == list(sfi[:n]), u
        flagged = want_doubt[u] or want_open[u]
        assert (u in redo) == flagged, (u, want_doubt[u], want_open[u])
        assert (u in realloc) == want_open[u], u
        assert (u in reana) == (flagged and not exact_flag[u]), u
        if not flagged:                                        # the bytes that are handed on without the exact kernels
            got = np.array(fld.q[:P.FIRST[n]])
            assert np.array_equal(got, want_q[u][:P.FIRST[n]]), (u, np.nonzero(got != want_q[u][:P.FIRST[n]])[0][:4])


@pytest.mark.parametrize('name,pcm', list(signals()), ids=[s[0] for s in signals()])
def test_scale_factor_guard_equals_its_model(ctx, name, pcm):
    from test_gpu_spec import spec_stages
    co, eps, side = spec_stages(ctx, [pcm])
    for f in range(co.shape[0]):
        slots = P.to_slots(co[f, 0])
        if not np.isfinite(eps[f, 0, :3]).all():
            assert eps[f, 0, 3].view(np.uint32) & 1                # bounds that are not finite: flagged
            continue
        sfi, unstable = P.sf_guard(slots, eps[f, 0, :3])
        assert np.array_equal(side[f, 0, :52], sfi), (f, np.nonzero(side[f, 0, :52] != sfi)[0][:4])
        assert bool(eps[f, 0, 3].view(np.uint32) & 1) == unstable, f
