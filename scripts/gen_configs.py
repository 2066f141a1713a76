#!/usr/bin/env python3
"""Writes the YAML configs of the three drivers.  Key names are the reference's; values are the
EFFECTIVE ones after PyYAML's last-duplicate-wins rule (several reference files define step_size_1 twice, e.g.
Tiny_ImageNet/configs_tinyimagenet/trades_training.yml:29,37 -> 1/255, not the 2/255 of the comment).
Keys the reference drivers read but some of its files omit (type_canny, step_size_3, n_queries) are always written."""
import os

ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "edge-enhancement_amd")
E255 = {1: 0.003921568627451, 2: 0.007843137254902, 16: 0.062745098039216}

ORDER = ["method_name", "arch", "start_epoch", "epochs", "batch_size", "lr", "momentum", "weight_decay", "workers", "pin_memory",
         "print_freq", "seed", "epsilon", "num_steps_1", "step_size_1", "num_steps_2", "step_size_2", "num_steps_3", "step_size_3",
         "random", "beta", "cize", "alpha", "sigma", "w", "r", "gf", "low", "high", "n_queries", "type_canny", "nGPU",
         "prob_start_from_clean", "label_smooth"]


def emit(path, header, d):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("# %s\n" % header)
        for k in ORDER:
            if k in d:
                v = d[k]
                if isinstance(v, bool):
                    v = "true" if v else "false"
                elif isinstance(v, str):
                    v = "'%s'" % v
                f.write("%s: %s\n" % (k, v))


def mnist(method, arch, **kw):
    d = dict(method_name=method, arch=arch, start_epoch=0, epochs=100, batch_size=50, lr=0.1, momentum=0.3, weight_decay=0.0001, workers=1,
             pin_memory=True, print_freq=100, seed=1, epsilon=0.3, num_steps_1=40, step_size_1=0.01, num_steps_2=50, step_size_2=0.01,
             num_steps_3=100, step_size_3=0.01, random=True, alpha=0, sigma=0, w=0, r=0, gf=False, low=0, high=0, n_queries=1,
             type_canny="None")
    d.update(kw)
    return d


def tiny(method, arch, step1, **kw):
    d = dict(method_name=method, arch=arch, start_epoch=0, epochs=50, batch_size=100, lr=0.1, momentum=0.9, weight_decay=0.0002, workers=2,
             pin_memory=True, print_freq=50, seed=1, epsilon=E255[16], num_steps_1=10, step_size_1=E255[step1], num_steps_2=50,
             step_size_2=E255[1], num_steps_3=100, step_size_3=E255[1], random=True, cize=64, alpha=0, sigma=0, w=0, r=0, gf=False, low=0,
             high=0, n_queries=1, type_canny="None")
    d.update(kw)
    return d


def imagenet(method, arch, step1, **kw):
    d = tiny(method, arch, step1, epochs=90, batch_size=256, weight_decay=0.0001, print_freq=100, cize=224, nGPU=4)
    d.update(kw)
    return d


EE_M = dict(alpha=0.3, sigma=1.0, w=1.0, r=4, low=25.0, high=51.0)
EE_T = dict(alpha=0, sigma=1.0, w=1.0, r=8, low=38.0, high=76.0)
EE_I = dict(alpha=0, sigma=1.0, w=1.0, r=16, low=38.0, high=76.0)
S125 = "CannyFilter_step125_1"

FILES = {
    "MNIST/configs_mnist/standard_training.yml": mnist("ST", "Net2"),
    "MNIST/configs_mnist/adversarial_training.yml": mnist("AT", "Net2"),
    "MNIST/configs_mnist/alp_training.yml": mnist("ALP", "Net2", beta=1.0),
    "MNIST/configs_mnist/trades_training.yml": mnist("TRADES", "Net2", lr=0.01, momentum=0.9, weight_decay=0, beta=1.0),
    "MNIST/configs_mnist/avmixup.yml": mnist("AVmixup", "Net2"),
    "MNIST/configs_mnist/ee_at_training.yml": mnist("EE_AT", "Net2_EE", type_canny="CannyFilter", **EE_M),
    "MNIST/configs_mnist/ee_at_bpda3_square.yml": mnist("EE_BPDA3_AT_square", "Net2_EE_square", type_canny=S125, **EE_M),
    "Tiny_ImageNet/configs_tinyimagenet/standard_training.yml": tiny("ST", "resnet18", 1),
    "Tiny_ImageNet/configs_tinyimagenet/adversarial_training.yml": tiny("AT", "resnet18", 2),
    "Tiny_ImageNet/configs_tinyimagenet/alp_training.yml": tiny("ALP", "resnet18", 1, beta=1.0),
    "Tiny_ImageNet/configs_tinyimagenet/trades_training.yml": tiny("TRADES", "resnet18", 1, beta=6.0),
    "Tiny_ImageNet/configs_tinyimagenet/avmixup_training.yml": tiny("AVmixup", "resnet18", 1),
    "Tiny_ImageNet/configs_tinyimagenet/ee_at_training.yml": tiny("EE_AT", "resnet18_EE", 1, type_canny="CannyFilter", **EE_T),
    "Tiny_ImageNet/configs_tinyimagenet/ee_at_square.yml": tiny("EE_AT_square", "resnet18_EE_square", 2, type_canny="CannyFilter", **EE_T),
    "Tiny_ImageNet/configs_tinyimagenet/ee_at_bpda3_square.yml": tiny("EE_BPDA3_AT_square", "resnet18_EE_square", 2, type_canny=S125, **EE_T),
    "Tiny_ImageNet/configs_tinyimagenet/ee_at_bpda3_pre_square.yml": tiny("EE_BPDA3_AT_pre_square", "resnet18_EE", 2, type_canny=S125, **EE_T),
    "Tiny_ImageNet/configs_tinyimagenet/processing_ee_at_square.yml": tiny("Processing_EE_AT_square", "resnet18_EE_square", 2,
                                                                            type_canny="CannyFilter", **EE_T),
    "Tiny_ImageNet/configs_tinyimagenet/targeted_adversarial_training.yml": tiny("tarAT", "resnet18", 1),
    "Tiny_ImageNet/configs_tinyimagenet/targeted_alp_training.yml": tiny("tarALP", "resnet18", 1, beta=1.0),
    "Tiny_ImageNet/configs_tinyimagenet/targeted_avmixup_training.yml": tiny("tarAVmixup", "resnet18", 1, beta=1.0),
    "Tiny_ImageNet/configs_tinyimagenet/targeted_ee_training.yml": tiny("tarEE", "resnet18_EE", 1, type_canny="CannyFilter", **EE_T),
    "Tiny_ImageNet/configs_tinyimagenet/targeted_ee_at_bpda3_square.yml": tiny("tarEE_BPDA3_AT_square", "resnet18_EE_square", 2,
                                                                                type_canny=S125, **EE_T),
    "ImageNet/configs_imagenet/standard_training.yml": imagenet("ST", "resnet18", 1),
    "ImageNet/configs_imagenet/advserarial_training.yml": imagenet("AT", "resnet18", 2),
    "ImageNet/configs_imagenet/targeted_advserarial_training.yml": imagenet("tarAT", "resnet18", 1),
    "ImageNet/configs_imagenet/targeted_alp_training.yml": imagenet("tarALP", "resnet18", 1, beta=1.0),
    "ImageNet/configs_imagenet/at_ee_training.yml": imagenet("EE_AT", "resnet18_EE", 1, type_canny=S125, **EE_I),
    "ImageNet/configs_imagenet/ee_at_bpda3_square.yml": imagenet("EE_AT_bpda3_square", "resnet18_EE_square", 1, type_canny=S125, **EE_I),
    "ImageNet/configs_imagenet/targeted_ee_training.yml": imagenet("tarEE", "resnet18_EE", 1, type_canny="CannyFilter", **EE_I),
    "ImageNet/configs_imagenet/targeted_ee_at_bpda3_square.yml": imagenet("tarEE_BPDA3_AT_square", "resnet18_EE_square", 2, type_canny=S125,
                                                                           **EE_I),
    "ImageNet/configs_imagenet/targeted_ee_trick_training.yml": imagenet("tarEE_trick", "resnet18_EE", 1, type_canny="CannyFilter",
                                                                          prob_start_from_clean=0.2, label_smooth=0.1, **EE_I),
}

if __name__ == "__main__":
    for rel, d in FILES.items():
        emit(os.path.join(ROOT, rel), "effective values of the reference's %s (scripts/gen_configs.py)" % rel, d)
    print("wrote %d configs" % len(FILES))
