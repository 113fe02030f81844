from .searcher import AVMNISTSearcher, MMIMDBSearcher, ModelSearcher, NTUSearcher  # noqa: F401
from .surrogate import SimpleRecurrentSurrogate, SurrogateDataloader, train_simple_surrogate  # noqa: F401
from . import tools  # noqa: F401
