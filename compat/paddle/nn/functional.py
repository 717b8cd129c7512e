import torch.nn.functional as _F

relu = _F.relu
softmax = _F.softmax
log_softmax = _F.log_softmax
