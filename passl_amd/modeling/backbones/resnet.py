"""ResNet backbone on the HIP path (depth 50/101/152 bottleneck blocks; depth 18/34 BasicBlock).

API and state_dict keys follow the reference's ``ResNet`` (passl_v110/modeling/backbones/
resnet.py:25-104, a subclass of paddle.vision's ResNet whose topology is stated in-tree at
passl_v110/modeling/backbones/resnetimagenet.py:111-253): ``ResNet(depth, num_classes=0,
with_pool=False, zero_init_residual=False, frozen_stages=-1, pretrained=None)``; stride on
conv2; bias-free convs; downsample = 1x1 conv(stride)+BN on the first block of a stage; init =
kaiming-normal(fan_out, relu) for convs, BN gamma=1 beta=0.

Execution differs from the reference by design: activations are NHWC in the compute dtype; in
training mode each conv is one implicit-GEMM kernel and BN+ReLU(+residual) three streaming
kernels; with frozen BN under no_grad (the key encoder) BN+ReLU+residual are folded into the conv
epilogue, i.e. one kernel per conv and no extra pass over the activations.
``forward`` takes the reference's NCHW fp32 image batch and returns the NHWC layer4 feature map.
"""
import torch

from ...hip import config, nn, ops, plan as P, streams
from ...modules import init
from ...utils.logger import get_logger
from ...modules.freeze import freeze_batchnorm_statictis
from .builder import BACKBONES


class BottleneckBlock(nn.Layer):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64,
                 dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2D
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = nn.Conv2D(inplanes, width, 1, bias_attr=False)
        self.bn1 = norm_layer(width)
        self.conv2 = nn.Conv2D(width, width, 3, padding=dilation, stride=stride, groups=groups,
                               dilation=dilation, bias_attr=False)
        self.bn2 = norm_layer(width)
        self.conv3 = nn.Conv2D(width, planes * self.expansion, 1, bias_attr=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU()
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # x forks into conv1 and the identity/downsample branch: the branch gradient is handed to
        # conv1's data-gradient kernel through a GradSlot (added in its epilogue) instead of an
        # autograd add kernel.
        slot = None
        if config.fuse_residual_grad() and torch.is_grad_enabled() and x.requires_grad:
            slot = nn.GradSlot()
            slot.arm()
        # conv1's data-gradient launch yields the COMPLETE gradient of x only when the identity /
        # downsample branch is folded in through the slot: then it may also do the backward reduction
        # of the BatchNorm that produced x (nn.BNLink); conv2 / conv3 are sole consumers of bn1 / bn2
        # the downsample branch (conv + BatchNorm of x) is independent of conv1..conv3: with the side
        # stream on it is issued there (AFTER the main branch on the host, so that its backward nodes
        # still run first and fill the GradSlot; on the GPU it starts as soon as x is complete) and
        # joined before bn3.  Autograd runs its backward on the side stream as well.
        # (not while the step is being captured into a HIP graph: ending a capture in which autograd ran a branch's
        # backward on the side stream by stream affinity crashes this ROCm build — profiles/r03_negative_results.txt;
        # the weight-gradient side stream and the frozen encoder's fork, which are ordered by our own events, capture fine)
        fork = (self.downsample is not None and torch.is_grad_enabled() and x.requires_grad and
                streams.enabled(x) and config.fork_downsample() and not torch.cuda.is_current_stream_capturing())
        x_ready = streams.record_event(torch.cuda.current_stream(x.device)) if fork else None
        out, st = self.conv1(x, want_stats=True, add_slot=slot,
                             producer=nn.bn_link(x) if slot is not None else None)
        out = self.bn1(out, relu=True, stats=st)
        out, st = self.conv2(out, want_stats=True, producer=nn.bn_link(out))
        out = self.bn2(out, relu=True, stats=st)
        out, st3 = self.conv3(out, want_stats=True, producer=nn.bn_link(out))
        if self.downsample is not None:
            if fork:
                main = torch.cuda.current_stream(x.device)
                side = streams.fork_stream(x.device)
                streams.wait_event(side, x_ready)
                with torch.cuda.stream(side):
                    idn, st = self.downsample[0](x, want_stats=True, sink_slot=slot)
                    identity = self.downsample[1](idn, relu=False, stats=st)
                streams.wait_event(main, streams.record_event(side))
                # `identity` comes from the side stream's pool and is read by bn3 on the main stream.  No
                # record_stream (hip/streams.py, "Memory"): every later piece of side-stream work starts with
                # a wait on a main-stream event recorded after this point, so the pool's stream-ordered
                # reuse is already behind bn3
            else:
                idn, st = self.downsample[0](x, want_stats=True, sink_slot=slot)
                identity = self.downsample[1](idn, relu=False, stats=st)
            return self.bn3(out, residual=identity, relu=True, stats=st3, res_link=nn.bn_link(identity))
        return self.bn3(out, residual=x, relu=True, stats=st3, res_slot=slot)   # out += identity; relu

    def forward_frozen(self, x, allow_fork=True):
        """Same block with running-stat BN folded into the conv epilogues (3-4 kernels).  allow_fork=False: the
        caller runs on a stream of its own (MoCo's key pipeline): the side stream's memory hand-off is ordered
        against the MAIN stream only (hip/streams.py), so a branch forked from a third stream could see its output
        recycled under it."""
        fork = allow_fork and self.downsample is not None and streams.enabled(x) and config.fork_downsample()
        x_ready = streams.record_event(torch.cuda.current_stream(x.device)) if fork else None
        out = self.conv1.infer(x, self.bn1, relu=True)
        out = self.conv2.infer(out, self.bn2, relu=True)
        identity = x
        if fork:
            # downsample conv next to conv1 / conv2 on the side stream; joined before conv3 (whose
            # epilogue adds it), so x outlives the side stream's reads
            main, side = torch.cuda.current_stream(x.device), streams.fork_stream(x.device)
            streams.wait_event(side, x_ready)
            with torch.cuda.stream(side):
                identity = self.downsample[0].infer(x, self.downsample[1], relu=False)
            streams.wait_event(main, streams.record_event(side))     # (no record_stream: see forward())
        elif self.downsample is not None:
            identity = self.downsample[0].infer(x, self.downsample[1], relu=False)
        return self.conv3.infer(out, self.bn3, residual=identity, relu=True)


class BasicBlock(nn.Layer):
    """The two-convolution block of depth 18 / 34 (reference passl_v110/modeling/backbones/resnetcifar.py:41-118):
    3x3(stride) -> BN -> ReLU -> 3x3 -> BN -> (+ downsample(x)) -> ReLU, bias-free convolutions.

    Same execution scheme as BottleneckBlock: every convolution is one implicit-GEMM launch with the BatchNorm
    statistics in its epilogue; the data-gradient launch of conv2 also reduces bn1's backward statistics (BNLink).
    conv1 is a 3x3 here, so the residual-fork hand-off (GradSlot: the identity gradient added in conv1's
    data-gradient epilogue, which then also reduces the previous BatchNorm's backward) is taken only when that
    data gradient is ONE dense launch — stride 1; a stride-2 conv1 (first block of stages 2-4, always next to a
    downsample branch) writes four sub-lattices of dx and leaves the sum of the two branch gradients to autograd."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64,
                 dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2D
        if dilation > 1:
            raise NotImplementedError('Dilation > 1 not supported in BasicBlock')
        self.conv1 = nn.Conv2D(inplanes, planes, 3, stride=stride, padding=1, bias_attr=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2D(planes, planes, 3, padding=1, bias_attr=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        slot = None
        if (config.fuse_residual_grad() and torch.is_grad_enabled() and x.requires_grad and self.stride == 1):
            slot = nn.GradSlot()
            slot.arm()
        out, st = self.conv1(x, want_stats=True, add_slot=slot,
                             producer=nn.bn_link(x) if slot is not None else None)
        out = self.bn1(out, relu=True, stats=st)
        out, st2 = self.conv2(out, want_stats=True, producer=nn.bn_link(out))
        if self.downsample is not None:
            idn, st = self.downsample[0](x, want_stats=True, sink_slot=slot)
            identity = self.downsample[1](idn, relu=False, stats=st)
            return self.bn2(out, residual=identity, relu=True, stats=st2)
        return self.bn2(out, residual=x, relu=True, stats=st2, res_slot=slot)

    def forward_frozen(self, x, allow_fork=True):
        """Running-statistics BatchNorm folded into the convolution epilogues: 2-3 launches per block."""
        out = self.conv1.infer(x, self.bn1, relu=True)
        identity = x if self.downsample is None else self.downsample[0].infer(x, self.downsample[1], relu=False)
        return self.conv2.infer(out, self.bn2, residual=identity, relu=True)


class _StagedInput:
    """An image batch already converted by ResNet.stage_input (tensor, size, completion event)."""
    __slots__ = ('xp', 'H', 'W', 'ready')

    def __init__(self, xp, H, W, ready):
        self.xp, self.H, self.W, self.ready = xp, H, W, ready


@BACKBONES.register()
class ResNet(nn.Layer):
    stem_pool = True          # ResNetsimclr (resnetcifar.py:275) drops the stem max-pool

    def __init__(self, depth, num_classes=0, with_pool=False, zero_init_residual=False,
                 frozen_stages=-1, pretrained=None):
        super().__init__()
        layer_cfg = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3], 50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}
        if depth not in layer_cfg:
            raise NotImplementedError('ResNet depth %r (built: 18 / 34 BasicBlock, 50 / 101 / 152 bottleneck)' % (depth,))
        block = BasicBlock if depth in (18, 34) else BottleneckBlock
        if num_classes > 0:
            raise NotImplementedError('classification fc is outside the MoCo hot path')
        layers = layer_cfg[depth]
        self.num_classes = num_classes
        self.with_pool = with_pool
        self._norm_layer = nn.BatchNorm2D
        self.inplanes = 64
        self.dilation = 1
        self.conv1 = nn.Conv2D(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias_attr=False)
        self.bn1 = self._norm_layer(self.inplanes)
        self.relu = nn.ReLU()
        if self.stem_pool:
            self.maxpool = nn.MaxPool2D(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        if with_pool:
            self.avgpool = nn.AdaptiveAvgPool2D((1, 1))
        self.zero_init_residual = zero_init_residual
        self.frozen_stages = frozen_stages
        self.init_parameters()
        if pretrained is not None:
            # paddle.load of a paddle.save'd dict = the reference's pickle-of-numpy checkpoint layout
            from ...utils.checkpoint import load_pickle, load_lenient
            state_dict = load_pickle(pretrained)
            if 'state_dict' in state_dict:
                state_dict = state_dict['state_dict']
            # lenient with warnings, like Layer.set_state_dict in the reference: published .pdparams
            # carry fc.* keys this trunk (num_classes=0) does not have
            load_lenient(self, state_dict, get_logger(), what='pretrained backbone')
            get_logger().info('Load pretrained backbone weight from {} success!'.format(pretrained))
        self._freeze_stages()

    def _make_layer(self, block, planes, blocks, stride=1):
        norm_layer = self._norm_layer
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = torch.nn.Sequential(
                nn.Conv2D(self.inplanes, planes * block.expansion, 1, stride=stride, bias_attr=False),
                norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, 1, 64, 1, norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, norm_layer=norm_layer))
        return torch.nn.Sequential(*layers)

    def init_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2D):
                init.kaiming_init(m, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn._BatchNormBase):
                init.constant_init(m, 1)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, BottleneckBlock):
                    init.constant_init(m.bn3, 0)
                elif isinstance(m, BasicBlock):
                    init.constant_init(m.bn2, 0)

    def _freeze_stages(self):
        """resnet.py:90-106: ``frozen_stages >= 0`` freezes the stem (conv1 / bn1), every further unit
        layer1..layer<frozen_stages>: their parameters are not trainable and their BatchNorms use the
        running statistics.  4 = the whole trunk (linear-probe configs, configs/moco/moco_clas_r50.yaml).
        The frozen prefix always runs the fused inference path (BatchNorm + ReLU + residual folded into
        the conv epilogues, one kernel per conv, nothing saved for backward)."""
        self.fully_frozen = False
        if self.frozen_stages < 0:
            return
        for m in self.frozen_modules():
            freeze_batchnorm_statictis(m)
            for p in m.parameters():
                p.requires_grad_(False)
        self.fully_frozen = self.frozen_stages >= 4
        get_logger().info('Frozen layer before stage {}'.format(self.frozen_stages + 1))

    def frozen_modules(self):
        """Sub-layers _freeze_stages freezes, in forward order."""
        if self.frozen_stages < 0:
            return []
        return [self.conv1, self.bn1] + [getattr(self, 'layer%d' % i) for i in range(1, min(self.frozen_stages, 4) + 1)]

    def trainable_modules(self):
        return [getattr(self, 'layer%d' % i) for i in range(max(self.frozen_stages, 0) + 1, 5)] \
            if self.frozen_stages >= 0 else [self.conv1, self.bn1, self.layer1, self.layer2, self.layer3, self.layer4]

    def _all_bn_frozen(self):
        return all(m.uses_global_stats() for m in self.modules() if isinstance(m, nn._BatchNormBase))

    def _stem_input(self, x):
        rt = nn._need_rt(self.conv1)
        _, _, H, W = x.shape
        _Hp, Wp = P.stem_padded_hw(H, W)
        return ops.nchw_to_nhwc_pad(x.contiguous().float(), P.STEM_PAD, Wp, P.STEM_CP, rt.arena.dtype), H, W

    def stage_input(self, x):
        """Layout conversion of an image batch (NCHW fp32 -> zero-padded NHWC in the compute dtype) ahead
        of time, on the side stream: MoCo converts the key view while the query encoder runs.  Returns
        what ``forward`` accepts in place of x (x itself when the side stream is off)."""
        if not streams.enabled(x):
            return x
        main, side = torch.cuda.current_stream(x.device), streams.side_stream(x.device)
        streams.wait_event(side, streams.record_event(main))
        with torch.cuda.stream(side):
            xp, H, W = self._stem_input(x)
        return _StagedInput(xp, H, W, streams.record_event(side))

    def units(self):
        """The trunk as a list of pipeline units: unit 0 = stem conv + BatchNorm (+ max-pool) + the first bottleneck,
        then one bottleneck per unit (16 units for depth 50).  -> [[sub-layers of unit u]]"""
        blocks = [b for st in (self.layer1, self.layer2, self.layer3, self.layer4) for b in st]
        return [[self.bn1, blocks[0]]] + [[b] for b in blocks[1:]]

    @torch.no_grad()
    def frozen_unit(self, u, x, allow_fork=True):
        """Unit u of the fused inference path on the CURRENT stream (see units()): unit 0 takes the image batch (or
        its staged form), every other unit the previous unit's output; the average pool follows the last unit.
        forward() of a fully frozen trunk is these calls in a row."""
        blocks = [b for st in (self.layer1, self.layer2, self.layer3, self.layer4) for b in st]
        if u == 0:
            if isinstance(x, _StagedInput):
                xp, H, W = x.xp, x.H, x.W
                streams.wait_event(torch.cuda.current_stream(xp.device), x.ready)
            else:
                xp, H, W = self._stem_input(x)
            y = self.conv1.infer(xp, self.bn1, relu=True, hw=(H, W))
            if self.stem_pool:
                y = self.maxpool(y)
        else:
            y = x
        y = blocks[u].forward_frozen(y, allow_fork)
        if u == len(blocks) - 1 and self.with_pool:
            y = self.avgpool(y)
        return y

    def forward(self, x):
        """x: [N,3,H,W] fp32 (reference layout) -> [N,H/32,W/32,2048] NHWC in the compute dtype."""
        if isinstance(x, _StagedInput):
            xp, H, W = x.xp, x.H, x.W
            main = torch.cuda.current_stream(xp.device)
            streams.wait_event(main, x.ready)      # xp: side-stream pool, read by the stem conv here (no record_stream: see BottleneckBlock.forward)
        else:
            xp, H, W = self._stem_input(x)
        stages = (self.layer1, self.layer2, self.layer3, self.layer4)
        # number of leading stages that run frozen: all of them for a key encoder under no_grad (every
        # BatchNorm on running statistics), the _freeze_stages prefix otherwise
        if self._all_bn_frozen() and (self.fully_frozen or not torch.is_grad_enabled()):
            n_frozen = 4
        else:
            n_frozen = min(self.frozen_stages, 4)
        if n_frozen >= 0:
            with torch.no_grad():
                y = self.conv1.infer(xp, self.bn1, relu=True, hw=(H, W))
                if self.stem_pool:
                    y = self.maxpool(y)
                for stage in stages[:n_frozen]:
                    for blk in stage:
                        y = blk.forward_frozen(y)
        else:
            y, st = self.conv1(xp, hw=(H, W), want_stats=True)
            if self.stem_pool:
                y = nn.bn_relu_maxpool(self.bn1, y, stats=st)     # one pass per direction (csrc/stem_pool.hip)
            else:
                y = self.bn1(y, relu=True, stats=st)
        # unit_done(u): called on the host right after unit u (units(): stem + first bottleneck, then one bottleneck
        # each) was enqueued — MoCo's key pipeline issues the key encoder's same unit from there
        unit_done = getattr(self, '_unit_done', None)
        u = 0
        for si, stage in enumerate(stages):
            for blk in stage:
                if si >= max(n_frozen, 0):
                    y = blk(y)
                    if unit_done is not None:
                        unit_done(u)
                u += 1
        if self.with_pool:
            y = self.avgpool(y)
        return y
