// placeholder, replaced below
#pragma once
