"""Host-side mirror of the reference's `my_package` Layer/Module wrappers
(SURVEY.md section 8 row W1).  Same class names, constructor arguments and
allocation semantics; written against today's torch.autograd.Function API."""
