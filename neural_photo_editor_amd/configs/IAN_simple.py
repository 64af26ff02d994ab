# IAN_simple model config in the reference's config-file format (a Python file exposing ``cfg`` and
# ``get_model``; loaded by API.py:18 / neural_photo_editor_amd.api.IAN).  Same graph, layer names and
# hyper-parameters as the reference's IAN_simple.py:33-51,56-241, written table-driven.
import lasagne
from lasagne.layers import batch_norm as BN, DenseLayer as DL, SliceLayer as SL
from lasagne.layers import TransposedConv2DLayer as TC2D, ReshapeLayer, InputLayer, GlobalPoolLayer
from lasagne.init import Normal
from lasagne.nonlinearities import elu, rectify as relu, tanh, sigmoid, LeakyRectify as lrelu
from layers import GaussianSampleLayer, MinibatchLayer

cfg = dict(batch_size=128, learning_rate={0: 0.0002}, optimizer='Adam', beta1=0.5, update_ratio=1, decay_rate=0,
           reg=1e-5, momentum=0.9, shuffle=True, dims=(64, 64), n_channels=3, n_classes=10, batches_per_chunk=64,
           max_epochs=250, checkpoint_every_nth=1, num_latents=100, recon_weight=3.0, feature_weight=1.0)

ENC_WIDTHS = (128, 256, 512, 1024)   # four 5x5 stride-2 convolutions, 64 -> 4 pixels
DEC_WIDTHS = (512, 256, 128)         # three 5x5 stride-2 transposed convolutions, 4 -> 32 pixels


def get_model(dnn=True):
    if dnn:
        from lasagne.layers.dnn import Conv2DDNNLayer as C2D
        from layers import DeconvLayer
    else:
        from lasagne.layers import Conv2DLayer as C2D
    conv = dict(filter_size=[5, 5], stride=[2, 2], W=Normal(0.02))
    net = l_in = InputLayer(shape=(None, cfg['n_channels']) + tuple(cfg['dims']))
    introspect = []
    for i, width in enumerate(ENC_WIDTHS):
        net = C2D(incoming=net, num_filters=width, pad=(2, 2), nonlinearity=lrelu(0.2), flip_filters=False,
                  name='enc_conv%d' % (i + 1), **conv)
        if i > 0:
            net = BN(net, name='bnorm%d' % (i + 1))
        introspect.append(net)
    enc_top = net
    fc1 = BN(DL(incoming=enc_top, num_units=1000, W=Normal(0.02), nonlinearity=elu, name='enc_fc1'), name='bnorm_enc_fc1')
    l_mu = BN(DL(incoming=fc1, num_units=cfg['num_latents'], nonlinearity=None, name='enc_mu'), name='mu_bnorm')
    l_ls = BN(DL(incoming=fc1, num_units=cfg['num_latents'], nonlinearity=None, name='enc_logsigma'), name='ls_bnorm')
    l_Z = GaussianSampleLayer(l_mu, l_ls, name='l_Z')
    net = BN(DL(incoming=l_Z, num_units=1024 * 16, nonlinearity=relu, W=Normal(0.02), name='l_dec_fc2'), name='bnorm_dec_fc2')
    net = ReshapeLayer(incoming=net, shape=([0], 1024, 4, 4))

    def up(incoming, width, name, nonlinearity, **kw):
        if dnn:
            return DeconvLayer(incoming=incoming, num_filters=width, crop=(2, 2), nonlinearity=nonlinearity, name=name, **dict(conv, **kw))
        return TC2D(incoming=incoming, num_filters=width, crop=(1, 1), nonlinearity=nonlinearity, name=name, **dict(conv, **kw))

    def trim(layer):  # the non-cuDNN transposed conv yields 2n+1 pixels: drop the first row / column
        return layer if dnn else SL(SL(layer, indices=slice(1, None), axis=2), indices=slice(1, None), axis=3)

    for i, width in enumerate(DEC_WIDTHS):
        net = trim(BN(up(net, width, 'dec_conv%d' % (i + 1), relu), name='bnorm_dc%d' % (i + 1)))
    l_out = trim(up(net, 3, 'dec_out', tanh, b=None))
    minibatch = MinibatchLayer(GlobalPoolLayer(enc_top), num_kernels=500, name='minibatch_discrim')
    l_discrim = DL(incoming=minibatch, num_units=1, nonlinearity=sigmoid, b=None, W=Normal(), name='discrimi')
    return {'l_in': l_in, 'l_out': l_out, 'l_mu': l_mu, 'l_ls': l_ls, 'l_Z': l_Z, 'l_introspect': introspect,
            'l_discrim': l_discrim}
