import sys, time, torch, cProfile, pstats
sys.path.insert(0, ".")
from seghiero_amd import ops, head as H
aspp = H.DepthwiseSeparableASPPModule(dilations=(1, 12, 24, 36), in_channels=2048, channels=512).to("cuda:0").train()
c4 = ops.new_act(16, 2048, 16, 16, torch.device("cuda:0")); c4.normal_().relu_()
cat = ops.new_act(16, 2560, 16, 16, torch.device("cuda:0"))
def unit():
    R = {}
    assert H._aspp_branches_grouped(aspp, c4, cat, 512, True, R)
for _ in range(3): unit()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): unit()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host issue us/unit", (t1 - t0) / 20 * 1e6, " wall incl. drain", (t2 - t0) / 20 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): unit()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
