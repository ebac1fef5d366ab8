import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from matryodshka_amd import MSI
from tests.util import make_inputs
m = MSI()
for (b,h,w,d) in ((1,320,640,32),(2,64,128,8)):
    inp = make_inputs(8964, b, h, w)
    ref = m.preprocess_image(torch.from_numpy(np.ascontiguousarray(inp["ref_image"])))
    src = m.preprocess_image(torch.from_numpy(np.ascontiguousarray(inp["src_image"])))
    planes = m.inv_depths(1.0, 100.0, d)
    pose = inp["src_pose"].copy(); pose[:, 0, 3] = 0.013; pose[:, 2, 3] = -0.02
    res = {}
    for ns in ("1", "2", "4"):
        os.environ["MSI_SWEEP_NS"] = ns
        psv = m.format_network_input(ref, src, inp["ref_pose"], pose, planes, inp["intrinsics"])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            m.format_network_input(ref, src, inp["ref_pose"], pose, planes, inp["intrinsics"])
        e1.record(); torch.cuda.synchronize()
        res[ns] = (psv.clone(), e0.elapsed_time(e1) / 20)
    print((b,h,w,d), "NS=1 %.4f ms  NS=2 %.4f ms  NS=4 %.4f ms  bitwise equal: %s %s" % (res["1"][1], res["2"][1], res["4"][1], torch.equal(res["1"][0], res["2"][0]), torch.equal(res["1"][0], res["4"][0])))
