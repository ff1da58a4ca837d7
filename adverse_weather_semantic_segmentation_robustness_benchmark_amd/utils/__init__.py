from .config import (Config, load_config, save_config, create_default_config, get_device_config,  # noqa: F401
                     setup_logging, validate_config)
