from .fast2q import main

if __name__ == "__main__":
    main()
