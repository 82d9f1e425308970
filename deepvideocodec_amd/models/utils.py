"""State-dict helpers and the straight-through quantiser of the DMC path.

Counterparts in the reference: dmc/models/utils.py:62-125 (buffer resizing before
`load_state_dict`) and :149-152 (`quantize_ste`).
"""
import torch


def resize_registered_buffers(module, module_name, buffer_names, state_dict):
    """Give the (initially empty) CDF buffers of `module` the sizes found in `state_dict`
    so that nn.Module.load_state_dict can copy into them.  Buffers that are already
    populated keep their size (a size mismatch is then reported by load_state_dict)."""
    own = dict(module.named_buffers())
    for name in buffer_names:
        if name not in own:
            raise ValueError(f'Invalid buffer name "{name}"')
    for name in buffer_names:
        key = f"{module_name}.{name}"
        if key not in state_dict:
            continue
        buf = own[name]
        if buf.numel() == 0:
            buf.resize_(state_dict[key].size())


# the reference's name for the same helper
update_registered_buffers = resize_registered_buffers
