"""Host time of one backbone call (enqueue only) against its GPU time, for plan variants."""
import os
import os.path as osp
import sys
import time

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
import torch  # noqa: E402


def main():
    import __graft_entry__ as ge
    from shapy_amd.utils import synthetic as syn
    net, _ = ge.make_network()
    bb = net.backbone
    x = torch.from_numpy(syn.synthetic_images(64, 224, 1)).cuda()
    for dag in (False, True, False, True):
        bb.dag = dag
        with torch.no_grad():
            for _ in range(3):
                bb(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                bb(x)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(f'dag={dag}: host enqueue {1e2 * (t1 - t0):.2f} ms per forward, total {1e2 * (t2 - t0):.2f} ms per forward')


if __name__ == '__main__':
    main()
