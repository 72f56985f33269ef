"""packages/utils.py of the reference (count_parameters :1-2, get_key :4-7)."""


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def get_key(my_dict, val):
    for key, value in my_dict.items():
        if val == value:
            return key
