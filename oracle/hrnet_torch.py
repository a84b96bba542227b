"""ORACLE (test infrastructure only -- never imported by shapy_amd/).

Plain PyTorch fp32 CPU restatement of the SHAPY HRNet-W48 backbone forward pass, written
functionally over a reference-layout ``state_dict`` (no nn.Module tree), so that the HIP
path can be compared with it on identical weights.

Follows regressor/human_shape/models/backbone/hrnet.py:
  * stem                      hrnet.py:427-432
  * layer1 (4 Bottlenecks)    hrnet.py:433, 342-359
  * transitions               hrnet.py:301-340, 435-462
  * HighResolutionModule      hrnet.py:175-193 (branches), 115-170 (fuse layers)
  * head                      hrnet.py:477-486 (subsample_*, concat, conv_layers, mean)
torchvision 0.8.2 BasicBlock / Bottleneck (third-party, not under /root/reference;
call sites hrnet.py:13,196-199,369-370) are restated from their published semantics:
BasicBlock = conv3x3-BN-ReLU-conv3x3-BN (+identity/downsample) ReLU; Bottleneck =
conv1x1-BN-ReLU-conv3x3-BN-ReLU-conv1x1-BN (+identity/downsample) ReLU.

Pinned by tests/golden/hrnet_golden.npz (outputs of the real reference module run on CPU
with the same seeded weights; generator: tests/golden/make_golden.py).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5

W48 = dict(
    stage2=dict(num_modules=1, num_branches=2, num_blocks=(4, 4), num_channels=(48, 96)),
    stage3=dict(num_modules=4, num_branches=3, num_blocks=(4, 4, 4),
                num_channels=(48, 96, 192)),
    stage4=dict(num_modules=3, num_branches=4, num_blocks=(4, 4, 4, 4),
                num_channels=(48, 96, 192, 384)),
)


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, BN_EPS)


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, padding)


def basic_block(sd, p, x):
    out = F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x, 1, 1)))
    out = _bn(sd, p + '.bn2', _conv(sd, p + '.conv2', out, 1, 1))
    return F.relu(out + x)


def bottleneck(sd, p, x, downsample):
    """downsample in {None, 'conv_bn' (layer1.0: Sequential(conv,bn)), 'conv' (conv_layers:
    a bare 1x1 conv, hrnet.py:367-370)}."""
    out = F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x)))
    out = F.relu(_bn(sd, p + '.bn2', _conv(sd, p + '.conv2', out, 1, 1)))
    out = _bn(sd, p + '.bn3', _conv(sd, p + '.conv3', out))
    if downsample == 'conv_bn':
        idt = _bn(sd, p + '.downsample.1', _conv(sd, p + '.downsample.0', x))
    elif downsample == 'conv':
        idt = _conv(sd, p + '.downsample', x)
    else:
        idt = x
    return F.relu(out + idt)


def hr_module(sd, p, xs, num_blocks):
    """HighResolutionModule.forward (hrnet.py:175-193)."""
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for b in range(num_blocks[i]):
            xs[i] = basic_block(sd, f'{p}.branches.{i}.{b}', xs[i])
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            q = f'{p}.fuse_layers.{i}.{j}'
            if j == i:
                t = xs[j]
            elif j > i:
                t = _bn(sd, q + '.1', _conv(sd, q + '.0', xs[j]))
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode='nearest')
            else:
                t = xs[j]
                for k in range(i - j):
                    t = _bn(sd, f'{q}.{k}.1', _conv(sd, f'{q}.{k}.0', t, 2, 1))
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_forward(sd, x, prefix='', return_stages=False):
    """x: [B,3,H,W] f32 (H, W multiples of 32).  Returns 'concat' features [B,2048]."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)} if prefix else sd
    stages = {}
    x = F.relu(_bn(sd, 'bn1', _conv(sd, 'conv1', x, 2, 1)))
    x = F.relu(_bn(sd, 'bn2', _conv(sd, 'conv2', x, 2, 1)))
    stages['stem'] = x
    for b in range(4):
        x = bottleneck(sd, f'layer1.{b}', x, 'conv_bn' if b == 0 else None)
    stages['layer1'] = x
    # transition1 (hrnet.py:435-440): both entries are conv3x3(+s2)+BN+ReLU of layer1
    xs = [F.relu(_bn(sd, 'transition1.0.1', _conv(sd, 'transition1.0.0', x, 1, 1))),
          F.relu(_bn(sd, 'transition1.1.0.1', _conv(sd, 'transition1.1.0.0', x, 2, 1)))]
    ys = hr_module(sd, 'stage2.0', xs, W48['stage2']['num_blocks'])
    stages['stage2'] = ys
    # transition2: new branch from y[-1] (hrnet.py:443-451)
    xs = ys + [F.relu(_bn(sd, 'transition2.2.0.1', _conv(sd, 'transition2.2.0.0', ys[-1], 2, 1)))]
    for m in range(W48['stage3']['num_modules']):
        xs = hr_module(sd, f'stage3.{m}', xs, W48['stage3']['num_blocks'])
    ys = xs
    stages['stage3'] = ys
    xs = ys + [F.relu(_bn(sd, 'transition3.3.0.1', _conv(sd, 'transition3.3.0.0', ys[-1], 2, 1)))]
    for m in range(W48['stage4']['num_modules']):
        xs = hr_module(sd, f'stage4.{m}', xs, W48['stage4']['num_blocks'])
    ys = xs
    stages['stage4'] = ys

    def subsample(name, t, n):
        for i in range(n):
            t = F.relu(_bn(sd, f'{name}.{3 * i + 1}', _conv(sd, f'{name}.{3 * i}', t, 2, 1)))
        return t
    feats = [subsample('subsample_4', ys[0], 3), subsample('subsample_3', ys[1], 2),
             subsample('subsample_2', ys[2], 1), ys[3]]
    xf = torch.cat(feats, dim=1)
    stages['cat'] = xf
    for b in range(5):
        xf = bottleneck(sd, f'conv_layers.{b}', xf, 'conv')
    stages['conv_layers'] = xf
    out = xf.mean(dim=(2, 3))
    if return_stages:
        return out, stages
    return out


def state_dict_spec():
    """(name, shape) of every HRNet-W48 (SHAPY variant) state_dict entry, in the reference's
    naming -- lets tests/bench build seeded weights without instantiating any module."""
    spec = []

    def conv(p, cin, cout, k, bias=False):
        spec.append((p + '.weight', (cout, cin, k, k)))
        if bias:
            spec.append((p + '.bias', (cout,)))

    def bn(p, c):
        spec.extend([(p + '.weight', (c,)), (p + '.bias', (c,)),
                     (p + '.running_mean', (c,)), (p + '.running_var', (c,)),
                     (p + '.num_batches_tracked', ())])

    conv('conv1', 3, 64, 3); bn('bn1', 64)
    conv('conv2', 64, 64, 3); bn('bn2', 64)

    def bottle(p, cin, planes, down):
        conv(p + '.conv1', cin, planes, 1); bn(p + '.bn1', planes)
        conv(p + '.conv2', planes, planes, 3); bn(p + '.bn2', planes)
        conv(p + '.conv3', planes, planes * 4, 1); bn(p + '.bn3', planes * 4)
        if down == 'conv_bn':
            conv(p + '.downsample.0', cin, planes * 4, 1); bn(p + '.downsample.1', planes * 4)
        elif down == 'conv':
            conv(p + '.downsample', cin, planes * 4, 1)

    bottle('layer1.0', 64, 64, 'conv_bn')
    for b in range(1, 4):
        bottle(f'layer1.{b}', 256, 64, None)
    conv('transition1.0.0', 256, 48, 3); bn('transition1.0.1', 48)
    conv('transition1.1.0.0', 256, 96, 3); bn('transition1.1.0.1', 96)

    def module(p, chans, nblocks):
        nb = len(chans)
        for i in range(nb):
            for b in range(nblocks[i]):
                q = f'{p}.branches.{i}.{b}'
                conv(q + '.conv1', chans[i], chans[i], 3); bn(q + '.bn1', chans[i])
                conv(q + '.conv2', chans[i], chans[i], 3); bn(q + '.bn2', chans[i])
        for i in range(nb):
            for j in range(nb):
                q = f'{p}.fuse_layers.{i}.{j}'
                if j > i:
                    conv(q + '.0', chans[j], chans[i], 1); bn(q + '.1', chans[i])
                elif j < i:
                    for k in range(i - j):
                        cout = chans[i] if k == i - j - 1 else chans[j]
                        conv(f'{q}.{k}.0', chans[j], cout, 3); bn(f'{q}.{k}.1', cout)

    module('stage2.0', (48, 96), (4, 4))
    conv('transition2.2.0.0', 96, 192, 3); bn('transition2.2.0.1', 192)
    for m in range(4):
        module(f'stage3.{m}', (48, 96, 192), (4, 4, 4))
    conv('transition3.3.0.0', 192, 384, 3); bn('transition3.3.0.1', 384)
    for m in range(3):
        module(f'stage4.{m}', (48, 96, 192, 384), (4, 4, 4, 4))

    def sub(name, cin, n):
        for i in range(n):
            conv(f'{name}.{3 * i}', cin, 2 * cin, 3, bias=True); bn(f'{name}.{3 * i + 1}', 2 * cin)
            cin *= 2
    sub('subsample_4', 48, 3)
    sub('subsample_3', 96, 2)
    sub('subsample_2', 192, 1)
    cin = 1536
    for b in range(5):
        bottle(f'conv_layers.{b}', cin, 512, 'conv')
        cin = 2048
    return spec
