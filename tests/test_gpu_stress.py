"""Long-run determinism stress of the LayerNorm statistics (the event it guards against: DESIGN.md section 4, "the wobble").

Round 4 saw, in ~0.1 % of back-to-back forwards of msi_train_net, one wave's share of a conv-transpose layer's sum of squares come out low (every stored value
bit-identical): a 1e-5 wobble of the consumer's normalisation, i.e. a different bit pattern in every later layer.  Round 5 traced it to one compiler-generated
packed-fp32 instruction of the generic epilogue (matryodshka_amd/isa_lint.py) and removed the instruction form from the library.  This test is the detector that
found it, bounded for the suite: thousands of forwards queued back to back, every prediction must equal the first bit for bit -- at the shape and on the kernels
the event was seen with (msi_train_net, batch 4, 128 x 256: ragged conv-transpose grids 17 x 37, 33 x 69, 65 x 133, three workgroups per CU on the fp16 form),
on both split arithmetics and both hand-off forms, plus CoordNet.  (Integer LayerNorm sums: any lost share changes the output bits of every later layer, so
comparing the final prediction covers every layer's statistics; tools/wobble_hunt.py compares the sums themselves and names the layer.)
r05 measurements of the same loop: before the fix 10-18 events per 6 000-10 000 forwards, after it 0 of 60 000."""
import pytest

from tests.test_gpu_cnn import env  # noqa: F401

pytestmark = pytest.mark.gpu
ALL = 0x3ffff


@pytest.mark.parametrize("coord,f16,fixup,shape,runs", [
    pytest.param(False, ALL, 0, (4, 128, 256, 48, 16, 64), 6000, id="wrapnet-f16x3-inlaunch"),
    pytest.param(False, ALL, 1, (4, 128, 256, 48, 16, 64), 3000, id="wrapnet-f16x3-fixup"),
    pytest.param(False, 0, 0, (4, 128, 256, 48, 16, 64), 3000, id="wrapnet-bf16x6"),
    pytest.param(True, ALL, 0, (4, 128, 256, 48, 16, 64), 3000, id="coordnet-f16x3"),
    pytest.param(False, ALL, 0, (1, 320, 640, 192, 64, 64), 1500, id="wrapnet-f16x3-640x320"),
    pytest.param(False, 0, 0, (1, 320, 640, 192, 64, 64), 1500, id="wrapnet-bf16x6-640x320"),
])
def test_thousands_of_back_to_back_forwards_are_bit_identical(env, coord, f16, fixup, shape, runs):
    torch, MSI, nets, N, onets = env
    b, h, w, cin, nout, ngf = shape
    m = MSI(weights=onets.init_weights(cin, nout, ngf=ngf, coord_net=coord, seed=29, randomize_affine=True), coord_net=coord)
    m.net_options[N.NET_OPT_F32_SPLIT_F16] = f16
    m.net_options[N.NET_OPT_FIXUP_KERNEL] = fixup
    x = torch.rand((b, h, w, cin), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) * 2 - 1
    plan = m._plan(b, h, w, cin, nout, ngf)
    kern = [plan.layer_kernel(i)[0] for i in range(17)]
    if not coord:   # msi_train_net's VALID conv-transposes run the split halo kernel, in the requested form
        assert sum("convt_halo_x3_kernel<%d>" % (2 if f16 else 3) in k for k in kern) == 3, kern
    first = m.run_net(x, nout, ngf).clone()
    assert bool(torch.isfinite(first).all())
    bad = 0
    for i in range(runs // 3):
        outs = [m.run_net(x, nout, ngf).clone() for _ in range(3)]   # three queued back to back, then compared (the clones are stream-ordered copies)
        bad += sum(0 if torch.equal(o, first) else 1 for o in outs)
    assert m.network_status() == 0
    assert bad == 0, "%d of %d forwards differ from the first" % (bad, runs)


@pytest.mark.parametrize("shape,runs,expect", [
    pytest.param((1, 320, 640, 192, 64, 64), 1500, "config1", id="headline-coordnet-bf16x6-1x320x640"),
    pytest.param((4, 640, 1280, 192, 64, 64), 1500, "big", id="config3-grid-coordnet-bf16x6-4x640x1280"),
])
def test_the_headline_plan_and_a_configs3_grid_are_bit_identical_over_thousands_of_forwards(env, shape, runs, expect):
    """the detector above stressed msi_train_net and CoordNet on the fp16 form, not the DEFAULT plan of BASELINE configs[1] (CoordNet, six-product
    form, 1 x 320 x 640: conv_halo8_x3 / convt_halo8_x3 / conv_halo8_s2_x3 and the row-parity conv_halo_x3<3, ...> kernels, all new in r05) nor a configs[3] / [4] grid
    (every layer on whole 8-row tiles).  The kernel list is asserted, so the test cannot silently stress something else after a plan change."""
    torch, MSI, nets, N, onets = env
    from tests.test_gpu_bench_plans import F32_CONFIG1, F32_BIG_GRID
    b, h, w, cin, nout, ngf = shape
    m = MSI(weights=onets.init_weights(cin, nout, ngf=ngf, coord_net=True, seed=31, randomize_affine=True), coord_net=True)
    x = torch.rand((b, h, w, cin), device="cuda", generator=torch.Generator(device="cuda").manual_seed(7)) * 2 - 1
    plan = m._plan(b, h, w, cin, nout, ngf)
    kern = [plan.layer_kernel(i)[0] for i in range(17)]
    assert kern == (F32_CONFIG1 if expect == "config1" else F32_BIG_GRID), kern
    if expect == "config1":
        assert {"conv_halo8_x3_kernel<0, 3>", "conv_halo8_x3_kernel<1, 3>", "convt_halo8_x3_kernel", "conv_halo8_s2_x3_kernel<1>", "conv_halo_x3_kernel<3, 1, 3>"} <= set(kern)
    first = m.run_net(x, nout, ngf).clone()
    assert bool(torch.isfinite(first).all())
    bad = 0
    for i in range(runs // 3):
        outs = [m.run_net(x, nout, ngf).clone() for _ in range(3)]
        bad += sum(0 if torch.equal(o, first) else 1 for o in outs)
    assert m.network_status() == 0
    assert bad == 0, "%d of %d forwards differ from the first" % (bad, runs)


def test_lds_staged_sweep_soak_against_the_gather_kernel():
    """Round 6: tools/sweep_soak.py, bounded for the suite (90 launches per case instead of 600: profiles/r06_sweep_soak.txt) -- the LDS-staged sweep (patch double buffer, one
    block barrier per frame, box reduction, register prefetch) at configs[2] / configs[3]-shard shapes and on mixed poses, every volume bit-identical to the gather kernel's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "sweep_soak.py"), "--runs", "90"], cwd=root, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0 and p.stdout.count(": 0 of 90 volumes differ") == 4, p.stdout[-3000:]
