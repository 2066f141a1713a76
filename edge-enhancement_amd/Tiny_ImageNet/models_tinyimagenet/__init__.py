"""Tiny-ImageNet model zoo: 64x64 inputs, 200 classes, adaptive average pooling
(reference: Tiny_ImageNet/models_tinyimagenet/{resnet,resnet_EE,resnet_EE_square}.py)."""
from eeadv.models import ResNet, ResNet_EE, make_resnet, make_resnet_ee

__all__ = ["ResNet", "ResNet_EE"]


def _plain(depth):
    def build(pretrained=False, **kwargs):
        return make_resnet(depth, "tiny", pretrained, **kwargs)
    build.__name__ = "resnet%d" % depth
    return build


def _ee(depth, square):
    def build(pretrained=False, **kwargs):
        return make_resnet_ee(depth, "tiny", square, pretrained, **kwargs)
    build.__name__ = "resnet%d_EE%s" % (depth, "_square" if square else "")
    return build


for _d in (18, 34, 50, 101, 152):
    globals()["resnet%d" % _d] = _plain(_d)
    globals()["resnet%d_EE" % _d] = _ee(_d, False)
    globals()["resnet%d_EE_square" % _d] = _ee(_d, True)
    __all__ += ["resnet%d" % _d, "resnet%d_EE" % _d, "resnet%d_EE_square" % _d]
