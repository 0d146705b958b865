"""Model factory with the reference's contract (wesep/models/__init__.py:10-27)."""
from . import bsrnn, bsrnn_multi_optim, convtasnet, dpccn, tfgridnet


def get_model(model_name: str):
    if model_name.startswith("BSRNN_Feats"):
        raise NotImplementedError(f"{model_name}: research variant outside the built hot path (SURVEY.md section 8)")
    if model_name.startswith("BSRNN_Multi"):
        return getattr(bsrnn_multi_optim, model_name)
    if model_name.startswith("BSRNN"):
        return getattr(bsrnn, model_name)
    if model_name.startswith("ConvTasNet"):
        return getattr(convtasnet, model_name)
    if model_name.startswith("DPCCN"):
        return getattr(dpccn, model_name)
    if model_name.startswith("TFGridNet"):
        return getattr(tfgridnet, model_name)
    print(model_name + " not found !!!")
    exit(1)
