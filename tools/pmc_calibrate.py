"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS library's access patterns.

MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a
wide coalesced streaming read (16 B per lane over whole kilobytes); "other access widths and
WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  The
conv kernels read 16 bytes per lane but in 64..128-byte row segments per pixel, so the factor
has to be measured: three launches whose HBM traffic is known a priori (tensors far larger than
the 256 MB Infinity Cache, each input byte needed once):

  1. direct  1x1  64 -> 64   on [64,256,256,64]   (1.07 GB in, 1.07 GB out)
  2. direct  3x3  48 -> 48   on [256,112,112,48]  (0.62 GB in, 0.62 GB out)
  3. Winograd 3x3 48 -> 48   on the same tensors

    rocprofv3 --pmc FETCH_SIZE  -d out -- python tools/pmc_calibrate.py
    rocprofv3 --pmc WRITE_SIZE  -d out -- python tools/pmc_calibrate.py
    python tools/pmc_calibrate.py --parse out      # -> JSON with reported / expected ratios
"""
import ctypes
import csv
import glob
import json
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))

CASES = [('direct_1x1_64', 64, 256, 256, 64, 64, 1, False),
         ('direct_3x3_48', 256, 112, 112, 48, 48, 3, False),
         ('winograd_3x3_48', 256, 112, 112, 48, 48, 3, True)]


def expected(case):
    _, B, H, W, Cin, Cout, ks, _ = case
    return {'read': 4.0 * B * H * W * Cin + 4.0 * Cout * ks * ks * Cin, 'write': 4.0 * B * H * W * Cout}


def run():
    import torch
    from shapy_amd import _lib
    from shapy_amd.utils import winograd
    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, B, H, W, Cin, Cout, ks, wino in CASES:
        x = torch.randn(B, H, W, Cin, device='cuda')
        w = torch.randn(Cout, ks, ks, Cin, device='cuda') * 0.05
        b = torch.randn(Cout, device='cuda')
        out = torch.empty(B, H, W, Cout, device='cuda')
        d = _lib.ShapyConv()
        d.in_ = x.data_ptr(); d.wgt = w.data_ptr(); d.bias = b.data_ptr(); d.out = out.data_ptr()
        d.B, d.Hi, d.Wi, d.Cin, d.in_ld = B, H, W, Cin, Cin
        d.Ho, d.Wo, d.Cout = H, W, Cout
        d.ksize, d.stride, d.pad = ks, 1, ks // 2
        d.out_ld = Cout; d.relu = 0; d.ups = 1; d.tile = 0 if wino else 0x2000
        d.dtype = _lib.DTYPE_F32
        if wino:
            wu = torch.from_numpy(winograd.transform_filters(w.cpu().numpy())).cuda()
            d.wgt_wino = wu.data_ptr()
        rc = lib.shapy_conv2d(ctypes.byref(d), stream)
        assert rc == 0, (name, rc)
        torch.cuda.synchronize()
        del x, w, out
        torch.cuda.empty_cache()


def parse(folder):
    rows = []
    for f in glob.glob(osp.join(folder, '**', '*counter_collection.csv'), recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if 'conv_' in r['Kernel_Name']]
    res = {'note': __doc__.split('\n\n')[0], 'cases': {}}
    for counter, key in (('FETCH_SIZE', 'read'), ('WRITE_SIZE', 'write')):
        rs = sorted([r for r in rows if r['Counter_Name'] == counter], key=lambda r: int(r['Dispatch_Id']))
        for case, r in zip(CASES, rs):
            exp = expected(case)[key]
            rep = float(r['Counter_Value']) * 1024.0
            res['cases'].setdefault(case[0], {})[key] = {
                'kernel': r['Kernel_Name'][:60], 'reported_bytes': rep, 'expected_bytes': exp,
                'reported_over_expected': rep / exp}
    print(json.dumps(res, indent=1))
    return res


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--parse':
        parse(sys.argv[2])
    else:
        run()
