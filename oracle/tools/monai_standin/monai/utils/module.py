from . import look_up_option, optional_import  # noqa: F401
