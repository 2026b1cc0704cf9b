"""timm.models stand-in: register_model (identity), named_apply (oracle tooling only)."""


def register_model(fn):
    return fn


def named_apply(fn, module, name="", depth_first=True, include_root=False):
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child in module.named_children():
        full = ".".join((name, child_name)) if name else child_name
        named_apply(fn=fn, module=child, name=full, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module
