"""Convert a reference-era checkpoint to a plain state_dict file.

The reference saves whole pickled module objects: `torch.save({'core_module': <models.unlg_former.Pansharpening>,
'iter_num': n})` (models/base/base_model.py:354-369), so its checkpoints can only be unpickled where the reference's
classes are importable -- i.e. in the build container with /root/reference (this tool uses tools/_ref_import.py).
The output is a reference-free file: {'core_module': OrderedDict(name -> tensor), 'iter_num': n} that
`lgteun_amd` loads anywhere:  runner.load_checkpoint(path)  /  module.load_state_dict(ckpt['core_module']).

    python tools/convert_checkpoint.py model_iter_255000.pth model_iter_255000.state.pth
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def convert(src, dst):
    from _ref_import import import_reference
    import_reference()                      # makes models.unlg_former.* resolvable for the unpickler
    ckpt = torch.load(src, map_location='cpu', weights_only=False)
    out = {}
    for k, v in ckpt.items():
        if hasattr(v, 'state_dict'):
            v = v.module if hasattr(v, 'module') else v          # nn.DataParallel wrapper (base_model.py:363-366)
            out[k] = {n: t.detach().clone() for n, t in v.state_dict().items()}
        else:
            out[k] = v
    torch.save(out, dst)
    return out


if __name__ == '__main__':
    if len(sys.argv) != 3:
        print(__doc__)
        sys.exit(2)
    o = convert(sys.argv[1], sys.argv[2])
    print({k: (len(v) if isinstance(v, dict) else v) for k, v in o.items()})
