# Full IAN model config (MDC blocks, RGB-Beta head, randomized IAF) in the reference's config-file format.
# Same graph, layer names and hyper-parameters as the reference's IAN.py:38-62,67-228, written table-driven.
import lasagne
from lasagne.layers import batch_norm as BN, DenseLayer as DL, SliceLayer as SL, ConcatLayer as CL
from lasagne.layers import ElemwiseSumLayer as ESL, NonlinearityLayer as NL, ReshapeLayer, InputLayer, GlobalPoolLayer
from lasagne.layers.dnn import Conv2DDNNLayer as C2D
from lasagne.init import Normal
from lasagne.nonlinearities import rectify as relu, sigmoid, softmax, LeakyRectify as lrelu
from layers import MDBLOCK, DeconvLayer, MinibatchLayer, beta_layer, MADE, IAFLayer, GaussianSampleLayer, MDCL

cfg = dict(batch_size=16, learning_rate={0: 0.0002, 25: 0.0001, 50: 0.00005, 75: 0.00001}, optimizer='Adam', beta1=0.5,
           update_ratio=1, decay_rate=0, reg=1e-5, momentum=0.9, shuffle=True, dims=(64, 64), n_channels=3,
           batches_per_chunk=64, max_epochs=80, checkpoint_every_nth=1, num_latents=100, recon_weight=3.0,
           feature_weight=1.0, dg_weight=1.0, dd_weight=1.0, agr_weight=1.0, ags_weight=1.0, n_shuffles=1, ortho=1e-3)

ENC_WIDTHS = (128, 256, 512, 1024)
# (deconv name, width, MDC block name, block scales): deconv -> residual MDC block, 4 -> 32 pixels
DEC_STAGES = (('dec_conv1', 512, 'dec_conv2a', [0, 2]), ('dec_conv2', 256, 'dec_conv3a', [0, 2, 3]),
              ('dec_conv3', 128, 'dec_conv4a', [0, 2, 3]))
HEAD_SCALES = [2, 3, 4]


def get_model(interp=False):
    conv = dict(filter_size=[5, 5], stride=[2, 2], W=Normal(0.02))
    net = l_in = InputLayer(shape=(None, cfg['n_channels']) + tuple(cfg['dims']))
    introspect = []
    for i, width in enumerate(ENC_WIDTHS):
        net = C2D(incoming=net, num_filters=width, pad=(2, 2), nonlinearity=lrelu(0.2), name='enc_conv%d' % (i + 1), **conv)
        if i > 0:
            net = BN(net, name='bnorm%d' % (i + 1))
        introspect.append(net)
    enc_top = net
    fc1 = BN(DL(incoming=enc_top, num_units=1000, W=Normal(0.02), nonlinearity=relu, name='enc_fc1'), name='bnorm_enc_fc1')
    l_mu = BN(DL(incoming=fc1, num_units=cfg['num_latents'], nonlinearity=None, name='enc_mu'), name='mu_bnorm')
    l_ls = BN(DL(incoming=fc1, num_units=cfg['num_latents'], nonlinearity=None, name='enc_logsigma'), name='ls_bnorm')
    l_Z_IAF = GaussianSampleLayer(l_mu, l_ls, name='l_Z_IAF')
    l_IAF_mu = MADE(l_Z_IAF, [cfg['num_latents']], 'l_IAF_mu')
    l_IAF_ls = MADE(l_Z_IAF, [cfg['num_latents']], 'l_IAF_ls')
    l_Z = IAFLayer(l_Z_IAF, l_IAF_mu, l_IAF_ls, name='l_Z')
    net = DL(incoming=l_Z, num_units=512 * 16, nonlinearity=lrelu(0.2), W=Normal(0.02), name='l_dec_fc2')
    net = ReshapeLayer(incoming=net, shape=([0], 512, 4, 4))
    for dc, width, block, scales in DEC_STAGES:
        net = DeconvLayer(incoming=net, num_filters=width, crop=(2, 2), nonlinearity=None, name=dc, **conv)
        net = MDBLOCK(incoming=net, num_filters=width, scales=scales, name=block, nonlinearity=lrelu(0.2))
    feat = BN(DeconvLayer(incoming=net, num_filters=128, crop=(2, 2), nonlinearity=lrelu(0.2), name='dec_conv4', **conv),
              name='bnorm_dc4')
    # autoregressive RGB-Beta head: each colour is a 2-channel (alpha, beta) map conditioned on the previous ones
    head = lambda src, name: MDCL(src, num_filters=2, scales=HEAD_SCALES, name=name)
    R = NL(head(feat, 'R'), sigmoid)
    G = NL(ESL([head(feat, 'G_a'), head(R, 'G_b')]), sigmoid)
    B = NL(ESL([head(feat, 'B_a'), head(CL([R, G]), 'B_b')]), sigmoid)
    l_out = CL([beta_layer(SL(c, slice(0, 1), 1), SL(c, slice(1, 2), 1)) for c in (R, G, B)])
    minibatch = MinibatchLayer(GlobalPoolLayer(enc_top), num_kernels=500, name='minibatch_discrim')
    l_discrim = DL(incoming=minibatch, num_units=3, nonlinearity=softmax, b=None, W=Normal(0.02), name='discrimi')
    return {'l_in': l_in, 'l_out': l_out, 'l_mu': l_mu, 'l_ls': l_ls, 'l_Z': l_Z, 'l_IAF_mu': l_IAF_mu,
            'l_IAF_ls': l_IAF_ls, 'l_Z_IAF': l_Z_IAF, 'l_introspect': introspect, 'l_discrim': l_discrim}
