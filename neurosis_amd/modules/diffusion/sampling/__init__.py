from .sigma_generators import DiscreteSigmaGenerator, EDMSigmaGenerator, InjectedSigmaGenerator, SigmaGenerator

__all__ = ["DiscreteSigmaGenerator", "EDMSigmaGenerator", "InjectedSigmaGenerator", "SigmaGenerator"]
