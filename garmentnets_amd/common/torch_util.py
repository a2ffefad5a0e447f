"""to_numpy -- /root/reference/common/torch_util.py:5-14"""


def to_numpy(x):
    return x.detach().to("cpu").numpy()
