def count_parameters(model) -> int:
    return int(sum(p.numel() for p in model.parameters(True)))
