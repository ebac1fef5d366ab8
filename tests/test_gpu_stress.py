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
